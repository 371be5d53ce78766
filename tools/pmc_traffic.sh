#!/bin/bash
# FETCH_SIZE and TCC hit/miss of kmer_grid_kernel at the bench's launch shape (separate --pmc passes) -> gpurun_out/pmc_traffic/
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
RPS=${RPS:-50000}
OUT=$ROOT/gpurun_out/pmc_traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $grp | cut -d' ' -f1)
    timeout -k 10 300 rocprofv3 --pmc $grp --kernel-include-regex kmer_grid --output-format csv -d $OUT/$name -- \
        python3 $ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --stage seeds --reads-per-step $RPS > $OUT/$name.log 2>&1
    echo "[pmc_traffic] $name done" | tee -a $OUT/progress.log
done
