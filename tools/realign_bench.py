"""End-to-end rate of the cpecan_realign command line on a BASELINE config-4-like input: two ~1 Mbp contigs related by
substitutions and short indels, N cigars of mixed lengths cut from their true alignment (20 % on the minus strand).
Writes the fasta and cigar files to a scratch directory, runs the binary once (one batch), prints cigars/s and the stage
times the library reports (CPECAN_REALIGN_TIMING).  Usage: python tools/realign_bench.py [N] [contig_bp]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGT", b"TGCA"):
    COMP[a] = b


def generate(n_cigars, contig):
    """Returns (directory holding seqs.fa and in.cigar, the cigar lines, aligned X bases)."""
    rng = np.random.default_rng(4)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    # the true alignment: match runs (geometric, mean 45) separated by indels of 1-6 bases of either kind
    ops, x_len = [], 0
    while x_len < contig:
        m = int(rng.geometric(1 / 45.0))
        ops.append((0, m))
        x_len += m
        k = int(rng.integers(1, 3))
        n = int(rng.integers(1, 7))
        ops.append((k, n))
        x_len += n if k == 1 else 0
    ops = np.array(ops[:-1], dtype=np.int64)  # ends with a match
    xs, ys = [], []
    for t, n in ops:
        seg = acgt[rng.integers(0, 4, n)]
        if t != 2:
            xs.append(seg)
        if t == 0:
            sub = seg.copy()
            hit = rng.random(n) < 0.1
            sub[hit] = acgt[rng.integers(0, 4, int(hit.sum()))]
            ys.append(sub)
        elif t == 2:
            ys.append(seg)
    X, Y = np.concatenate(xs), np.concatenate(ys)
    Yrc = COMP[Y[::-1]]
    x_at = np.concatenate([[0], np.cumsum(np.where(ops[:, 0] != 2, ops[:, 1], 0))])
    y_at = np.concatenate([[0], np.cumsum(np.where(ops[:, 0] != 1, ops[:, 1], 0))])
    d = tempfile.mkdtemp(prefix="realign_bench_")
    with open(os.path.join(d, "seqs.fa"), "w") as f:
        for name, s in (("contigX", X), ("contigY", Y), ("contigYrc", Yrc)):
            f.write(">%s\n%s\n" % (name, s.tobytes().decode()))
    op_char = "MDI"
    lines, bases = [], 0
    for _ in range(n_cigars):
        first = 2 * int(rng.integers(0, len(ops) // 2))
        want = int(rng.choice([150, 300, 600, 1200, 2400]))
        last = first
        while last + 2 < len(ops) and x_at[last + 1] - x_at[first] < want:
            last += 2
        x0, x1, y0, y1 = x_at[first], x_at[last + 1], y_at[first], y_at[last + 1]
        text = " ".join("%s %d" % (op_char[t], n) for t, n in ops[first:last + 1])
        if rng.random() < 0.2:
            lines.append("cigar: contigYrc %d %d - contigX %d %d + 100 %s" % (len(Y) - y0, len(Y) - y1, x0, x1, text))
        else:
            lines.append("cigar: contigY %d %d + contigX %d %d + 100 %s" % (y0, y1, x0, x1, text))
        bases += x1 - x0
    with open(os.path.join(d, "in.cigar"), "w") as f:
        f.write("\n".join(lines) + "\n")
    return d, lines, bases


def main():
    n_cigars = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    contig = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    d, lines, bases = generate(n_cigars, contig)
    cig = os.path.join(d, "in.cigar")
    exe = os.path.join(ROOT, "cpecan_amd", "cpecan_realign")
    env = dict(os.environ, CPECAN_REALIGN_TIMING="1")
    for label in ("cold", "warm"):
        t0 = time.time()
        with open(cig) as fin:
            res = subprocess.run([exe, "--batch", str(n_cigars), os.path.join(d, "seqs.fa")], stdin=fin, capture_output=True,
                                 text=True, env=env)
        dt = time.time() - t0
        assert res.returncode == 0, res.stderr[-2000:]
        out = [l for l in res.stdout.split("\n") if l]
        assert len(out) == n_cigars
        print("%s: %d cigars (%d aligned X bases) in %.2f s wall = %.0f cigars/s  [process start, HIP init, fasta + cigar "
              "text included]" % (label, n_cigars, bases, dt, n_cigars / dt))
        print("   " + res.stderr.strip().replace("\n", "\n   "))
    for batch in (1024, 4096, 16384):  # smaller batches: more launches and uploads for the same cigars
        t0 = time.time()
        with open(cig) as fin:
            res = subprocess.run([exe, "--batch", str(batch), os.path.join(d, "seqs.fa")], stdin=fin, capture_output=True,
                                 text=True)
        dt = time.time() - t0
        assert res.returncode == 0 and [l for l in res.stdout.split("\n") if l] == out
        print("--batch %d: %.2f s wall = %.0f cigars/s, same output" % (batch, dt, n_cigars / dt))
    same = sum(a.split()[10:] == b.split()[10:] for a, b in zip(lines, out))
    print("cigars whose operations are unchanged by the realignment: %d of %d" % (same, n_cigars))


if __name__ == "__main__":
    main()
