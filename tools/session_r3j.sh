set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3j; rm -rf $O; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "item_gather or abi_is_stateless" > $O/pytest_new.log 2>&1; echo "pytest new rc=$?"
tail -12 $O/pytest_new.log
