set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3d; rm -rf $O; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
timeout 120 ./tools/mb_bwd2.bin 320 32 800 > $O/mb_bwd2_cfg2.txt 2>&1; cat $O/mb_bwd2_cfg2.txt
timeout 60 ./tools/mb_bwd2.bin 384 8 400 > $O/mb_bwd2_refyaml.txt 2>&1; grep "us/step" $O/mb_bwd2_refyaml.txt
timeout 60 ./tools/mb_bwd2.bin 128 8 300 > $O/mb_bwd2_cfg1.txt 2>&1; grep "us/step" $O/mb_bwd2_cfg1.txt
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider -k "rnn or model_three or fused_dropout or side_stream or elementwise or shipped or large_shape" > $O/pytest_rnn.log 2>&1; echo "pytest rnn rc=$?"
tail -5 $O/pytest_rnn.log
