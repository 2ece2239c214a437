python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "instance_norm" 2>&1 | tail -2
for i in 1 2 3; do
python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', d['ms_per_step'])"
HDF_LIB_PATH=$PWD/.ab/prev.so python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prev', d['ms_per_step'])"
done
