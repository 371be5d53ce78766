# same-box A/B runs of the default bench under different switches: bash tools/sweep_r03.sh  (on the GPU box)
run() { name=$1; shift; env "$@" timeout -k 10 240 python bench.py --steps 3 --warmup 1 --cpu-seconds 1 $EXTRA > gpurun_out/sw_$name.json 2> gpurun_out/sw_$name.err || echo "$name failed"; python -c "import json; d=json.loads(open('gpurun_out/sw_$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],2), d['config']['stage_ms_per_step'])"; }
EXTRA="" run lds0 LRSC_WP_LEAVES_LDS=0 &&
EXTRA="" run lds1 LRSC_WP_LEAVES_LDS=1 &&
EXTRA="" run lds0b LRSC_WP_LEAVES_LDS=0 &&
EXTRA="" run lds1b LRSC_WP_LEAVES_LDS=1
