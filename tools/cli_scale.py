"""End-to-end run of the `stride` binary at a moderate scale: index + pbcorrect (default flow and --nodp), wall-clock times.
usage: python tools/cli_scale.py GENOME_MB N_READS [extra pbcorrect options]"""
import subprocess, sys, time, tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from longreadselfcorrect_amd import Lrsc

genome_mb, n_reads = float(sys.argv[1]), int(sys.argv[2])
more = sys.argv[3:]            # extra pbcorrect options, e.g. --devices 0,0 --batch 10000
api = Lrsc()
g = api.synth_genome(0x5EED0001, int(genome_mb * 1e6))
bases, off = api.synth_reads(0x5EED0002, g, n_reads, 10000)
tmp = Path(tempfile.mkdtemp(dir="/tmp"))
fa = tmp / "reads.fa"
buf = bases.tobytes()
with open(fa, "w") as f:
    for i in range(n_reads):
        f.write(f">r{i}\n{buf[int(off[i]): int(off[i + 1])].decode()}\n")
stride = str(Path(__file__).resolve().parents[1] / "longreadselfcorrect_amd" / "_build" / "stride")
mb = int(off[-1]) / 1e6
t = time.time()
subprocess.run([stride, "index", "-p", str(tmp / "idx"), str(fa)], check=True, capture_output=True)
print(f"stride index: {mb:.0f} Mbases in {time.time() - t:.1f}s", flush=True)
for extra in (["--nodp"], []):
    out = tmp / ("out_nodp" if extra else "out_dp")
    t = time.time()
    r = subprocess.run([stride, "pbcorrect", "-p", str(tmp / "idx"), "-o", str(out), "-c", "90", "-g", "5"] + extra + more + [str(fa)],
                       capture_output=True, text=True)
    dt = time.time() - t
    assert r.returncode == 0, r.stderr[-2000:]
    stats = {l.split(":")[0]: l.split(":")[1].strip() for l in r.stdout.strip().split("\n") if ":" in l}
    print(f"stride pbcorrect {' '.join(extra) or '(default)'}: {dt:.1f}s wall = {mb / dt:.1f} Mbases/s incl. index load, FASTA in/out; "
          f"walks {stats.get('TotalWalkNum')} FM {stats.get('FMNum')} DP {stats.get('DPNum')} corrected {stats.get('CorrectedLen')}", flush=True)
