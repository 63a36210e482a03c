cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1400 python -m pytest tests -q -m gpu 2>&1 | tail -3
timeout 600 python tools/soak_parity.py 1000000 2>&1 | grep "fp32" | cut -c1-200
for s in 21 22; do timeout 200 python tools/epa_staged_check.py 1000000 $s 2>&1 | tail -1; done
