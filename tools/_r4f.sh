out=gpurun_out/r4f; mkdir -p $out
timeout 2700 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 | tee $out/pytest.txt
