# round-3 session F: whole GPU suite after the stateless-ABI refactor + scatter2 policy, then the soak
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3f; rm -rf $O; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1800 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -s > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"
grep -E "^\[cfg|^\[ref" $O/pytest_all.log | head -20
tail -6 $O/pytest_all.log
timeout 1500 python tools/soak.py --out $O/r03_soak.json > $O/soak.out 2> $O/soak.err; echo "soak rc=$?"
tail -5 $O/soak.err
