run() { python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
run base
HDF_SMALL_TILE_MAX=32768 run small32k
HDF_TINY_TILE_MAX=4096 run tiny4k
HDF_SMALL_TILE_MAX=32768 HDF_TINY_TILE_MAX=4096 run both
run base
HDF_SMALL_TILE_MAX=32768 HDF_TINY_TILE_MAX=4096 run both
