#!/usr/bin/env python3
"""bench.py -- corrected-read throughput of the PacBio self-correction hot path on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Default workload = BASELINE.json configs[2]: 100k x 10 kb reads (15 % error) per GPU, corrected against the
FM-index of that same 90x read set (1.05 G symbols per strand = the "1 Gb FM-index"), default flow of
`stride pbcorrect -c 90 -g 5` (seeds -> seed-to-seed FM-extension -> DP/MSA fallback -> stitching), everything on
the device, corrected strings and counters downloaded to the host inside the timed region.

A "step" is one pass of that whole per-read path over one resident sub-batch of `--reads-per-step` reads (the
rank's reads are cut into sub-batches that stay in HBM; step i processes sub-batch i mod n).  `value` = input
bases of the timed steps / wall time (max over ranks), summed over ranks.  Reads shard across ranks, the read-only
index is replicated in every GPU's HBM; there is no collective on the data path, only
the timing barrier.

One JSON line on stdout (rank 0):
  roofline      the graded Occ-rank kernel (kmer_grid_kernel) of the same run: algorithmic bytes (rank blocks +
                k-mer table lines, 64 B each, counted on the device) / its HIP-event time on the ctx stream
  roofline_extra  the same pricing for the FM-extension kernels of the walk-parallel flow (wp_prepare / wp_begin / wp_extend: where
                the metric lives)
  cpu_baseline  the CPU oracle (port of the reference algorithm) on all host cores, one thread per core over
                disjoint reads of the first sub-batch, bounded to about --cpu-seconds
  parity_sample the oracle's corrected strings and integer counters for those sampled reads compared with what
                the GPU produced for the same reads in the timed run; a mismatch makes the process exit 3
`--stage seeds` times configs[1] (Occ-rank + LongReadProbe kernels only) and says so in `metric`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BLOCK_BYTES = 64

T0 = time.time()


def log(msg: str):
    print(f"[bench +{time.time() - T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


STAGE_INFO = {
    "seeds": dict(config=1, metric="seed-stage Mbases/s (whole node; Occ-rank + LongReadProbe kernels only, BASELINE configs[1])",
                  stages=["LongReadProbe k-mer feature grid (Occ-rank kernel)",
                          "getSeqAttribute + searchSeedsWithHybridKmers + estimateBestKmerSize + removeHitchhikingSeeds"]),
    "correct-nodp": dict(config=2, metric="corrected Mbases/s (whole node), --nodp flow",
                         stages=["seed stage", "seed-to-seed FM-extension of every seed pair at once + per-read stitching (wp_* kernels)",
                                 "download of corrected strings + counters"]),
    "correct": dict(config=2, metric="corrected Mbases/s (whole node)",
                    stages=["seed stage", "seed-to-seed FM-extension of every seed pair at once + per-read stitching (wp_* kernels)",
                            "DP/MSA fallback rounds (LF-walk retrieval, extendMatch, multiple alignment + consensus)",
                            "download of corrected strings + counters"]),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-mb", type=float, default=11.1, help="synthetic genome size; 11.1 Mb x 90x = 100k x 10 kb reads")
    ap.add_argument("--reads", type=int, default=100_000, help="reads per GPU (= index reads on rank 0)")
    ap.add_argument("--read-len", type=int, default=10_000)
    ap.add_argument("--reads-per-step", type=int, default=0,
                    help="reads of one step's resident sub-batch (0 = stage default: all reads for `seeds`; 50000 otherwise = half of the rank's shard, so that the driver's ""--steps 20 --warmup 5 fits its 600 s, see DESIGN.md section 4a)")
    ap.add_argument("--streams", type=int, default=0,
                    help="correct stages: a step's sub-batch is cut into this many parts, corrected concurrently by one host thread + "
                         "ctx (HIP stream) each, so that the DP rounds of one part overlap the extension of the others "
                         "(default 2 with GPU_MAX_HW_QUEUES=16: measured 32.6 vs 28.8 Mbases/s for 1 part; with the runtime's default of 4 "
                         "hardware queues the streams of the two parts queue behind each other's persistent kernels and 2 parts are slower; "
                         "3 and 4 parts are slower again: the persistent kernels contend for wavefront slots)")
    ap.add_argument("--no-stagger", action="store_true", help="start the concurrent parts of a step together instead of staggered")
    ap.add_argument("--stagger-on", choices=["align", "dp"], default="align",
                    help="what part j waits for in part j-1: its first extendMatch launch done (align), or its DP stage entered (dp: the "
                         "LF-walk launch in front of it)")
    ap.add_argument("--pipeline", action="store_true",
                    help="several concurrent parts: no join per step (see run_steps); measured 65.8 vs 59.9 corrected Mbases/s, with the Occ-rank "
                         "kernel's own time doubled by the contention")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (0 disables it and parity_sample)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="oracle threads of the cpu_baseline leg (0 = all host cores of this process)")
    ap.add_argument("--stage", choices=list(STAGE_INFO), default="correct",
                    help="correct (default): BASELINE configs[2], the whole per-read path with the DP fallback; correct-nodp: the "
                         "same with --nodp; seeds: configs[1], Occ-rank + LongReadProbe kernels only")
    args = ap.parse_args()
    sinfo = STAGE_INFO[args.stage]

    # The two concurrent parts of a step use ~10 HIP streams between them (main + yield + MSA side streams each); the ROCm runtime
    # multiplexes streams onto 4 hardware queues by default, where a part's short kernels queue behind the other part's persistent
    # correction kernel.  Must be set before the runtime initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback for the product path")
    # LRSC_BENCH_ONE_DEVICE=1: rehearsal of the multi-rank path on a one-GPU box (every rank on device 0, gloo for the barrier)
    one_device = os.environ.get("LRSC_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from longreadselfcorrect_amd import Lrsc
    from longreadselfcorrect_amd import dist as lrdist
    from longreadselfcorrect_amd.capi import K_GRID, K_SEEDS, K_EXTEND, K_LF, K_DP, K_MSA

    api = Lrsc()
    genome_len = int(args.genome_mb * 1e6)
    n_reads = args.reads

    # ---- setup (untimed): data, index, upload ---------------------------------------------------
    log(f"rank {rank}: synthetic genome {genome_len / 1e6:.1f} Mb, {n_reads} x {args.read_len} reads")
    genome = api.synth_genome(0x5EED0001, genome_len)
    idx_bases, idx_off = api.synth_reads(0x5EED0002, genome, n_reads, args.read_len, first_read=0)
    n_sym = int(idx_off[-1]) + n_reads
    log(f"index read set: {int(idx_off[-1]) / 1e6:.1f} Mbases, {n_sym / 1e9:.3f} G symbols per strand; building BWTs on the GPU")
    t = time.time()
    units = [api.build_bwt(idx_bases, idx_off, rev, local_rank) for rev in (False, True)]
    build_s = time.time() - t
    log(f"BWT + rBWT built in {build_s:.1f}s ({units[0].size / 1e6:.0f} M / {units[1].size / 1e6:.0f} M RL units)")
    t = time.time()
    index = api.index_from_units(units[0], units[1], n_reads, n_sym)
    index.upload(local_rank)
    info = index.info()
    upload_s = time.time() - t
    log(f"rank-block image + k-mer tables built and uploaded in {upload_s:.1f}s: {info.device_bytes / 1e9:.2f} GB of rank blocks in HBM, "
        f"{info.block_symbols} symbols per {info.block_bytes}-byte block")
    params = api.params_default(5, 90)          # -g 5 -c 90: k = 17, pool {5,9,15,17,19}
    params.no_dp = 1 if args.stage == "correct-nodp" else 0
    ctx = index.ctx(params, local_rank)

    if rank == 0:
        bases, off = idx_bases, idx_off          # self-correction: rank 0 corrects the indexed reads themselves
    else:
        bases, off = api.synth_reads(0x5EED0002, genome, n_reads, args.read_len,
                                     first_read=lrdist.weak_shard_first_read(rank, n_reads))
    per_step = args.reads_per_step or (n_reads if args.stage == "seeds" else 50_000)
    per_step = max(1, min(per_step, n_reads))
    n_streams = 1 if args.stage == "seeds" else max(1, args.streams or 2)
    cuts = list(range(0, n_reads, per_step)) + [n_reads]
    if len(cuts) > 2 and cuts[-1] - cuts[-2] < per_step // 2:      # fold a short last sub-batch into the one before it
        del cuts[-2]
    ctxs = [ctx] + [index.ctx(params, local_rank) for _ in range(n_streams - 1)]
    groups, group_bases = [], []          # one group per step: n_streams resident parts, one per ctx
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        bounds = [lo + (hi - lo) * j // n_streams for j in range(n_streams + 1)]
        parts = []
        for c, plo, phi in zip(ctxs, bounds[:-1], bounds[1:]):
            sub_off = (off[plo: phi + 1] - off[plo]).astype(np.uint64)
            parts.append(c.batch(bases[int(off[plo]): int(off[phi])], sub_off))
        groups.append(parts)
        group_bases.append(int(off[hi]) - int(off[lo]))
    batches = [b for g in groups for b in g]
    reads_per_step = sum(b.n_reads for b in groups[0])
    log(f"{len(groups)} sub-batch(es) of {reads_per_step} reads resident in HBM ({n_streams} concurrent part(s) each), "
        f"{int(off[-1]) / 1e6:.1f} Mbases in all")

    totals = {"walks": 0, "fm": 0, "dp": 0, "corrected_reads": 0, "corrected_bases": 0, "not_ok": 0}
    kept = {}            # results of part 0 of sub-batch 0 from its latest pass (parity_sample compares them with the oracle)

    def correct_part(g, j, timed, acc):
        b = groups[g][j]
        res, poff, out = b.correct()      # FM-extension chain (+ DP/MSA rounds) and stitching on the device; results on the host
        if timed:
            acc.append((sum(r.total_walk_num for r in res), sum(r.fm_num for r in res), sum(r.dp_num for r in res),
                        sum(1 for r in res if r.merge), int(out.size), sum(1 for r in res if r.status != 0)))
        if g == 0 and j == 0:
            kept["res"], kept["poff"], kept["out"] = res, poff.copy(), out.copy()

    def part_work(g, j, timed, acc):
        # one part of a step: its seed stage, then its correction
        groups[g][j].find_seeds()      # k-mer grid (Occ-rank kernel) + getSeqAttribute + greedy seed scan, all on the device
        if args.stage != "seeds":
            correct_part(g, j, timed, acc)

    def run_steps(first, count, timed):
        """`count` steps = count x n_streams parts.
        Default: per step, the seed stage of every part first, one after the other (each launch has the GPU to itself: the k-mer grid
        kernel's HIP-event time in this run is its own), then the parts' corrections concurrently, joined at the end of the step.
        --pipeline: each host thread (one per ctx) takes its own part of every step, one step after the other WITHOUT a join per step,
        so that the FM-extension of one part overlaps the DP stage of the other across steps (+10 % corrected Mbases/s); the Occ-rank
        kernel then shares the device with the other part's kernels and its own roofline figure halves -- not the default for that
        reason."""
        acc = []
        if n_streams > 1 and not args.pipeline:
            for i in range(first, first + count):
                g = i % len(groups)
                for b in groups[g]:
                    b.find_seeds()      # k-mer grid (Occ-rank kernel) + getSeqAttribute + greedy seed scan, all on the device
                if args.stage != "seeds":
                    # The parts of a step start STAGGERED: part j + 1 starts when part j has reached its DP stage (or is through), so that
                    # one part's FM-extension (latency-bound, persistent kernels) shares the device with another's DP stage (issue-bound)
                    # instead of both parts doing the same thing at the same time.
                    done = [threading.Event() for _ in range(n_streams)]

                    def staggered(j):
                        if j > 0 and args.stage == "correct" and not args.no_stagger:
                            kq = K_DP if args.stagger_on == "align" else K_LF
                            base = ctxs[j - 1].stats(kq).launches
                            while not done[j - 1].is_set() and ctxs[j - 1].stats(kq).launches == base:
                                time.sleep(0.005)
                        try:
                            correct_part(g, j, timed, acc)
                        finally:
                            done[j].set()

                    ths = [threading.Thread(target=staggered, args=(j,)) for j in range(n_streams)]
                    for th in ths: th.start()
                    for th in ths: th.join()
                if timed and rank == 0:
                    log(f"step {i - first + 1}/{count} done, {time.perf_counter() - t0:.1f}s into the timed region")
        elif n_streams == 1:
            for i in range(first, first + count):
                part_work(i % len(groups), 0, timed, acc)
                if timed and rank == 0:
                    log(f"step {i - first + 1}/{count} done, {time.perf_counter() - t0:.1f}s into the timed region")
        else:
            def worker(j):
                for i in range(first, first + count):
                    part_work(i % len(groups), j, timed, acc)
                    if timed and rank == 0 and j == 0:
                        log(f"step {i - first + 1}/{count}: part 0 done, {time.perf_counter() - t0:.1f}s into the timed region")
            ths = [threading.Thread(target=worker, args=(j,)) for j in range(n_streams)]
            for th in ths: th.start()
            for th in ths: th.join()
        for a_ in acc:
            for key, v in zip(("walks", "fm", "dp", "corrected_reads", "corrected_bases", "not_ok"), a_):
                totals[key] += v
        return sum(group_bases[i % len(groups)] for i in range(first, first + count))

    t0 = time.perf_counter()
    if args.warmup:
        t_w = time.perf_counter()
        run_steps(0, args.warmup, False)
        log(f"{args.warmup} warmup step(s): {time.perf_counter() - t_w:.1f}s")
    for c in ctxs:
        c.stats_reset()

    def fence():
        for c in ctxs:
            c.sync()
        torch.cuda.synchronize()
        lrdist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    my_bases = run_steps(args.warmup, args.steps, True)       # progress lines per step inside (watchdogs)
    fence()
    elapsed = time.perf_counter() - t0

    # max over ranks of the elapsed time; sum over ranks of the bases processed
    elapsed_max, total_bases = lrdist.combine(elapsed, float(my_bases), device="cpu" if one_device else "cuda")

    class Agg:
        def __init__(self, which):
            sts = [c.stats(which) for c in ctxs]
            self.launches = sum(x.launches for x in sts); self.total_ms = sum(x.total_ms for x in sts)
            self.block_loads = sum(x.block_loads for x in sts); self.table_loads = sum(x.table_loads for x in sts)
            self.rank_queries = sum(x.rank_queries for x in sts)

    def roofline_of(which, name):
        st = Agg(which)
        launches = max(st.launches, 1)
        ms = st.total_ms / launches
        lines = (st.block_loads + st.table_loads) / launches
        ach = lines * BLOCK_BYTES / max(ms * 1e-3, 1e-12) / 1e9
        return {"kernel": name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": None, "algorithmic_gb_per_launch": lines * BLOCK_BYTES / 1e9, "kernel_ms": ms, "launches": int(st.launches),
                "block_loads_per_launch": st.block_loads / launches, "table_loads_per_launch": st.table_loads / launches,
                "rank_queries_per_launch": st.rank_queries / launches}

    result = None
    rc = 0
    if rank == 0:
        roof = roofline_of(K_GRID, "kmer_grid_kernel")
        # HBM-side traffic of the same kernel on the same launch shape comes from the committed rocprofv3 --pmc passes (counters
        # cannot be sampled from inside the process); null for any other workload
        pmc = REPO / "profiles" / "r03_pmc" / "traffic.json"
        roof["peak_measured"] = measured_copy_peak(torch)
        roof["frac_of_measured"] = roof["achieved"] / roof["peak_measured"] if roof["peak_measured"] else None
        roof["peak_measured_how"] = "device-to-device copy of 4 GiB inside this run (read + write bytes / HIP-event time, best of 5)"
        if pmc.exists():
            pj = json.loads(pmc.read_text())
            # the counters were collected for one build of the kernel: only quote them for that build
            same_build = pj.get("kernel_source_sha256") == grid_kernel_source_hash()
            for shape in pj.get("shapes", [pj]):
                if same_build and shape.get("reads_per_launch") == batches[0].n_reads and shape.get("index_reads") == n_reads and shape.get("read_len") == args.read_len:
                    roof["traffic"] = shape["traffic_bytes_per_launch"] / 1e9
                    roof["traffic_unit"] = "GB per launch (separate rocprofv3 --pmc passes, corrected as MI355X_MICROARCH.md prescribes)"
                    roof["traffic_source"] = shape["source"]
        value = total_bases / elapsed_max / 1e6
        stage_ms = {"kmer_grid": Agg(K_GRID).total_ms / args.steps, "seed_scan_group": Agg(K_SEEDS).total_ms / args.steps}
        if args.stage != "seeds":
            # summed over the concurrent parts: with n_streams > 1 these overlap in time and add up to more than the step
            stage_ms.update({"fm_extend_and_stitch": Agg(K_EXTEND).total_ms / args.steps,
                             "fm_extend_launches_per_step": Agg(K_EXTEND).launches / args.steps,
                             "dp_retrieve_lf_walks": Agg(K_LF).total_ms / args.steps,
                             "dp_extend_match": Agg(K_DP).total_ms / args.steps,
                             "dp_msa_consensus": Agg(K_MSA).total_ms / args.steps,
                             "concurrent_parts": n_streams})
        result = {
            "metric": sinfo["metric"],
            "value": value,
            "unit": "Mbases/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int64",
            "data": "synthetic",
            "config": {
                "workload": (f"BASELINE configs[{sinfo['config']}]: {n_reads} x {args.read_len / 1000:g} kb reads/GPU (15% err: 4.5% del, 1.5% sub, "
                             f"9% ins) over the FM-index of the {n_reads}-read 90x set of a {args.genome_mb:g} Mb genome "
                             f"({n_sym / 1e9:.2f} G symbols/strand), -c 90 -g 5{' --nodp' if args.stage == 'correct-nodp' else ''}; "
                             f"one step = one resident sub-batch of {reads_per_step} reads ({group_bases[0] / 1e6:.0f} Mbases"
                             f"{', corrected as %d concurrent parts' % n_streams if n_streams > 1 else ''}) through "
                             f"{'the seed stage' if args.stage == 'seeds' else 'the whole per-read path, results downloaded'}"),
                "stage": args.stage,
                "stages_timed": sinfo["stages"],
                "stage_ms_per_step": stage_ms,
                "reads_per_step": reads_per_step,
                "sub_batches": len(groups),
                "concurrent_parts_per_step": n_streams,
                **({} if args.stage == "seeds" else {
                    "walks": totals["walks"], "fm_walks": totals["fm"], "dp_walks": totals["dp"],
                    "corrected_reads": totals["corrected_reads"], "corrected_bases_out": totals["corrected_bases"],
                    "reads_status_not_ok": totals["not_ok"]}),
                "index_hbm_gb": info.device_bytes / 1e9,
                "index_build_s": build_s, "index_upload_and_tables_s": upload_s,
                "reads_per_gpu": n_reads,
                "parallelism": f"reads sharded x{world}, index replicated, no data-path collective",
            },
            "roofline": roof,
        }
        if args.stage != "seeds":
            result["roofline_extra"] = [roofline_of(K_EXTEND, "wp_prepare + wp_begin + wp_extend kernels (one timed group per round)")]
        if world == 1 and args.cpu_seconds > 0:
            cb, ps = cpu_baseline(units, n_reads, n_sym, params, bases, off, batches[0].n_reads, args, kept, batches[0])
            result["cpu_baseline"] = cb
            if ps is not None:
                result["parity_sample"] = ps
                if not ps["identical"]:
                    rc = 3
        print(json.dumps(result), flush=True)

    for b in batches:
        b.close()
    for c in ctxs:
        c.close()
    index.close()
    if world > 1:
        dist.destroy_process_group()
    if rc:
        log("parity_sample: the GPU results differ from the CPU oracle's for the sampled reads")
        sys.exit(rc)


def grid_kernel_source_hash() -> str:
    """sha256 over the sources the Occ-rank kernel is compiled from (profiles/r03_pmc/traffic.json records the same digest)."""
    import hashlib

    h = hashlib.sha256()
    for f in ("kernels.hip", "rank_device.h", "fm_device.h"):
        h.update((REPO / "longreadselfcorrect_amd" / "csrc" / f).read_bytes())
    return h.hexdigest()


def measured_copy_peak(torch) -> float:
    """Stream-copy rate of this GPU in GB/s (SURVEY.md section 8d asks for it beside the nominal peak)."""
    try:
        n = 1 << 30                                   # 4 GiB of int32
        a = torch.empty(n, dtype=torch.int32, device="cuda")
        b = torch.empty_like(a)
        a.zero_()
        best = 0.0
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); b.copy_(a); e1.record(); torch.cuda.synchronize()
            best = max(best, 2 * 4 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        del a, b
        torch.cuda.empty_cache()
        return best
    except Exception:
        return 0.0


def host_cores() -> int:
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box hands a one-GPU job a
    16-core share of a 256-core host; 256 oracle threads on 16 cores would run 16x over the time budget)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    cap = int(os.environ.get("LRSC_BENCH_MAX_CPU_THREADS", "32"))
    return max(1, min(n, cap))


def cpu_baseline(units, n_reads, n_sym, params, bases, off, batch0_reads, args, kept, batch0):
    """The CPU oracle (port of the reference's path for the same stages) on a bounded sample of sub-batch 0, one thread per
    host core over disjoint reads (ctypes releases the GIL; the index is shared read-only).  For the correct stages the oracle's
    strings and counters of the sampled reads are compared with the GPU's (parity_sample)."""
    from oracle import oracle_py

    stage = args.stage
    threads = args.cpu_threads or host_cores()
    log(f"cpu_baseline: loading the index into the CPU oracle (RLBWT markers); {threads} thread(s)")
    orc = oracle_py.Oracle()
    ob = orc.bwt_from_units(units[0], n_reads, n_sym)
    orb = orc.bwt_from_units(units[1], n_reads, n_sym)

    def run_chunk(lo, hi):
        sub_off = (off[lo: hi + 1] - off[lo]).astype(np.uint64)
        sub = bases[int(off[lo]): int(off[hi])]
        if stage == "seeds":
            return orc.find_seeds(ob, orb, params, sub, sub_off)
        run = orc.correct_reads(ob, orb, params, sub, sub_off)
        got = (run.correct_fa, run.counters.copy())
        run.close()
        return got

    # calibrate on 2 reads (one thread), then size the sample to the budget
    t = time.perf_counter()
    run_chunk(0, 2)
    per_read = (time.perf_counter() - t) / 2
    n = int(max(threads, min(batch0_reads, args.cpu_seconds / max(per_read, 1e-6) * threads)))
    n -= n % threads
    log(f"cpu_baseline: {per_read * 1e3:.0f} ms/read/thread -> sampling {n} reads over {threads} threads")
    bounds = [n * i // threads for i in range(threads + 1)]
    outs = [None] * threads

    def worker(i):
        outs[i] = run_chunk(bounds[i], bounds[i + 1])

    t = time.perf_counter()
    ths = [threading.Thread(target=worker, args=(i,)) for i in range(threads)]
    for th in ths: th.start()
    for th in ths:
        while th.is_alive():
            th.join(timeout=30.0)
            if th.is_alive():
                log(f"cpu_baseline: {sum(o is not None for o in outs)} of {threads} oracle threads done")
    dt = time.perf_counter() - t
    sample_bases = int(off[n])
    cb = {"value": sample_bases / dt / 1e6, "unit": "Mbases/s", "cores": threads, "kind": "port",
          "per_core": sample_bases / dt / 1e6 / threads,
          "sample": f"first {n} reads ({sample_bases / 1e6:.2f} Mbases) of sub-batch 0, "
                    f"{'seed stage (LongReadProbe::searchSeedsWithHybridKmers)' if stage == 'seeds' else 'whole per-read path (PacBioSelfCorrectionProcess::process, no_dp=%d)' % params.no_dp}, "
                    f"oracle/ restatement over the reference's RLBWT layout (g++ -O3, no -march), {threads} threads x {n // threads} reads, {dt:.1f}s; "
                    f"the port runs at about 0.75x the reference binary's own per-core rate (0.030 vs 0.040 Mbases/s/core in BASELINE.md's "
                    f"survey run: the reference itself cannot be built in this image)"}
    if stage == "seeds":
        # compare the seeds of the sampled reads with the GPU's
        count_g, seeds_g, _ = batch0.seeds(want_attribute=False)
        ok, first_bad, pos = True, None, 0
        starts = np.concatenate([[0], np.cumsum(count_g.astype(np.int64))])
        for i in range(threads):
            count_o, seeds_o, _ = outs[i]
            lo, hi = bounds[i], bounds[i + 1]
            g = seeds_g[int(starts[lo]): int(starts[hi])]
            g_arr = np.stack([g[f] for f in g.dtype.names], axis=1) if g.size else np.zeros((0, 8), dtype=np.int32)
            if not (np.array_equal(count_o, count_g[lo:hi]) and np.array_equal(np.asarray(seeds_o).reshape(-1, 8), g_arr)):
                ok, first_bad = False, lo
                break
        return cb, {"reads": n, "identical": ok, "compared": "seed counts + all 8 fields of every seed", "first_mismatch_chunk_at_read": first_bad}
    if "res" not in kept:
        return cb, None
    res, poff, out = kept["res"], kept["poff"], kept["out"].tobytes()
    ok, first_bad = True, None
    fields = ("total_reads_len", "corrected_len", "total_seed_num", "total_walk_num", "high_error_num", "exceed_depth_num",
              "exceed_leave_num", "fm_num", "dp_num", "seed_dis")
    for i in range(threads):
        fa_o, ctr_o = outs[i]
        lo, hi = bounds[i], bounds[i + 1]
        parts = []
        for r in range(lo, hi):
            R = res[r]
            if R.merge:
                for pj in range(R.piece_first, R.piece_first + R.n_pieces):
                    parts.append(f">r{r - lo}\n{out[int(poff[pj]): int(poff[pj + 1])].decode()}\n")
        ctr_g = np.array([[getattr(res[r], f) for f in fields] + [1 if res[r].merge else 0] for r in range(lo, hi)], dtype=np.int64)
        if "".join(parts) != fa_o or not np.array_equal(ctr_g, ctr_o):
            ok, first_bad = False, lo
            break
    return cb, {"reads": n, "identical": ok, "compared": "correct.fa records + the 10 integer counters + merge flag of every sampled read",
                "first_mismatch_chunk_at_read": first_bad}


if __name__ == "__main__":
    main()
