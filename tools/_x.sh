cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for s in 1 2 3; do timeout 600 python tools/cfg4d_ids.py 100000 $s 2>&1 | tail -3; done
timeout 900 python tools/mesh_soak.py --seeds 6 --n 50000 2>&1 | tail -6
