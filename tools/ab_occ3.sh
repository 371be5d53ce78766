# on the GPU box: default build vs extension kernel compiled for 3 wavefronts per SIMD with the walk's pieces merely not `noinline`
# (record of a round-3 experiment against the build that still had the pieces as calls; wp.hip now defines the inlining itself, so
#  only -DLRSC_WP_EXTEND_OCC=3 is left to vary here)
set -e
B=longreadselfcorrect_amd/_build
run() { name=$1; shift; env "$@" timeout -k 10 240 python bench.py --steps 3 --warmup 1 --cpu-seconds 1 > gpurun_out/occ_$name.json 2> gpurun_out/occ_$name.err || echo "$name failed"; python -c "import json; d=json.loads(open('gpurun_out/occ_$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],2), d['config']['stage_ms_per_step'])"; }
run base A=1
cp $B/liblrsc_hip.so /tmp/liblrsc_base.so
rm -f $B/obj/wp.hip.o
make -C longreadselfcorrect_amd/csrc EXTRA="-DLRSC_WP_EXTEND_OCC=3" > gpurun_out/occ_build.log 2>&1
run occ3 LRSC_WP_LANES=196608
run occ3_lanes2 A=1
timeout -k 10 300 python -m pytest tests/test_gpu_fm.py -m gpu -x -q -k "whole_path_with_dp_fallback or repeat_dataset" > gpurun_out/occ_tests.log 2>&1; echo tests rc=$?; tail -2 gpurun_out/occ_tests.log
