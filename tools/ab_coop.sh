# on the GPU box: exactness of the run-ahead helper launch, then a same-box A/B of the default bench
LRSC_WP_COOP=1 timeout -k 10 240 python -m pytest tests/test_gpu_real_shape.py tests/test_gpu_fm.py -m gpu -x -q -k "10kb_reads_default_flow or 10kb_reads_nodp or capacity" > gpurun_out/coop_tests.log 2>&1; rc=$?; echo tests rc=$rc; tail -2 gpurun_out/coop_tests.log
[ $rc -eq 0 ] || exit 1
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-seconds 1 > gpurun_out/coop_$name.json 2> gpurun_out/coop_$name.err || { echo "$name failed"; return 1; }; python -c "import json; d=json.loads(open('gpurun_out/coop_$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],2), d['config']['stage_ms_per_step'])"; }
run off LRSC_WP_COOP=0 && run on LRSC_WP_COOP=1
