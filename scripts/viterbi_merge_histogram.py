"""Merge-depth histogram of consecutive tracebacks (round-5 verdict item 7: "traceback reuse") -- an analysis tool, CPU only.

viterbi_decode re-walks tb_depth - 2 hops from the first-minimum state of every trellis step (convcode.py:644-657).  The walk from
best(t) could stop as soon as it reaches a state the walk from best(t - 1) already visited: from there on both follow the same
survivor.  This script measures how soon that happens on BASELINE config 2 (K = 7 (133,171), 1024-bit blocks, QPSK + AWGN, soft
decisions) -- per codeword (one lane of the codeword-per-lane kernel) and per group of 64 codewords (one wavefront: the early exit
has to be wave-uniform, so a wave pays the LARGEST depth among its 64 lanes).

depth d(t) = the smallest d >= 1 with  anc_d(best(t)) == anc_{d-1}(best(t - 1))   (anc_d = d hops back along the survivors);
d = H + 1 where the two walks have not met within the H = tb_depth - 2 hops of a walk.

The forward pass below is a self-contained NumPy batch ACS (float64, the reference's operation order); it does not use oracle/.

usage: python scripts/viterbi_merge_histogram.py [--ebn0 3.0] [--codewords 4096] [--out profiles/r06_viterbi_merge_histogram.md]
"""
import argparse
import sys

import numpy as np

sys.path.insert(0, ".")
from commpy_amd.channelcoding import Trellis, conv_encode_batch          # noqa: E402  (host-side table builder and encoder only)


def forward(x, nxt, out, S, n_steps):
    """Batch add-compare-select: choice[t, b, s] (0 = even predecessor), best[t, b].  Shift-register trellis, k = 1, n = 2."""
    B = x.shape[0]
    ps = np.zeros((S, 2), np.int64)
    pc = np.zeros((S, 2), np.int64)
    cnt = np.zeros(S, np.int64)
    for p in range(S):
        for i in range(2):
            s = nxt[p, i]
            ps[s, cnt[s]], pc[s, cnt[s]] = p, out[p, i]
            cnt[s] += 1
    pm = np.full((B, S), np.inf)
    pm[:, 0] = 0.0
    choice = np.zeros((n_steps + 1, B, S), np.int8)
    best = np.zeros((n_steps + 1, B), np.int64)
    rows = np.arange(B)[:, None]
    n_llr = x.shape[1] // 2
    with np.errstate(over="ignore"):
        for t in range(1, n_steps + 1):
            r = x[:, 2 * (t - 1):2 * t] if t <= n_llr else np.zeros((B, 2))
            nll0 = np.log(np.exp(r) + 1)
            nll1 = nll0 - r
            bm = np.stack([nll0[:, 0] + nll0[:, 1], nll0[:, 0] + nll1[:, 1], nll1[:, 0] + nll0[:, 1], nll1[:, 0] + nll1[:, 1]], axis=1)
            cand = pm[:, ps] + bm[rows, pc.reshape(-1)].reshape(B, S, 2)
            ch = np.argmin(cand, axis=2)
            pm = np.take_along_axis(cand, ch[:, :, None], axis=2)[:, :, 0]
            choice[t] = ch
            best[t] = np.argmin(pm, axis=1)
    return ps, choice, best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ebn0", type=float, default=3.0)
    ap.add_argument("--codewords", type=int, default=4096)
    ap.add_argument("--seed", type=int, default=6)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    tr = Trellis(np.array([6]), np.array([[0o133, 0o171]]))
    S, m = tr.number_states, tr.total_memory
    L_msg, B = 1024, a.codewords
    rs = np.random.RandomState(a.seed)
    msg = rs.randint(0, 2, (B, L_msg))
    coded = conv_encode_batch(msg, tr)                                # [B, 2 * (1024 + 6)]
    sigma2 = 1.0 / (2 * 0.5 * 10 ** (a.ebn0 / 10.0))                  # per real dimension, QPSK with unit-energy bits, rate 1/2
    y = (2.0 * coded - 1.0) + rs.randn(*coded.shape) * np.sqrt(sigma2)
    llr = 2.0 * y / sigma2                                            # log P(1) / P(0)
    L = coded.shape[1] // 2
    tb = 5 * m
    H = tb - 2
    n_steps = L + m - 1
    ps, choice, best = forward(llr, np.asarray(tr.next_state_table), np.asarray(tr.output_table), S, n_steps)
    br = np.arange(B)
    # path[t][d] = anc_d(best(t)) for d = 0 .. H, all lanes at once
    depth = np.zeros((n_steps + 1, B), np.int64)
    prev_path = None
    for t in range(1, n_steps + 1):
        hops = min(H, t - 1)
        path = np.zeros((hops + 1, B), np.int64)
        st = best[t]
        path[0] = st
        for d in range(1, hops + 1):
            st = ps[st, choice[t - d + 1, br, st]]
            path[d] = st
        if prev_path is not None:
            dmax = min(hops, prev_path.shape[0])                      # compare anc_d(best(t)) with anc_{d-1}(best(t-1)), d = 1 .. dmax
            eq = path[1:dmax + 1] == prev_path[:dmax]
            first = np.where(eq.any(axis=0), eq.argmax(axis=0) + 1, H + 1)
            depth[t] = first
        prev_path = path
    d = depth[tb:]                                                    # steps with a full-length walk
    lane_hist = np.bincount(d.reshape(-1), minlength=H + 2)[1:]
    wave = d.reshape(d.shape[0], B // 64, 64).max(axis=2)
    wave_hist = np.bincount(wave.reshape(-1), minlength=H + 2)[1:]
    lines = []
    lines.append("# Merge depth of consecutive Viterbi tracebacks (config 2, Eb/N0 = %.1f dB, %d codewords = %d wavefronts, %d steps each)"
                 % (a.ebn0, B, B // 64, d.shape[0]))
    lines.append("")
    lines.append("`scripts/viterbi_merge_histogram.py --ebn0 %.1f --codewords %d --seed %d` (CPU, NumPy float64 forward pass; "
                 "walk length H = tb_depth - 2 = %d hops, `convcode.py:644-657`)." % (a.ebn0, B, a.seed, H))
    lines.append("")
    lines.append("depth = hops until the walk from best(t) meets the walk from best(t-1); %d = not met inside the walk." % (H + 1))
    lines.append("")
    lines.append("| depth | share of (codeword, step) | share of (wavefront, step): the largest of 64 lanes |")
    lines.append("|---|---|---|")
    lt, wt = lane_hist.sum(), wave_hist.sum()
    for i in range(H + 1):
        if lane_hist[i] or wave_hist[i]:
            lines.append("| %d | %.4f %% | %.4f %% |" % (i + 1, 100.0 * lane_hist[i] / lt, 100.0 * wave_hist[i] / wt))
    mean_lane = (lane_hist * np.arange(1, H + 2)).sum() / lt
    mean_wave = (wave_hist * np.arange(1, H + 2)).sum() / wt
    lines.append("")
    lines.append("mean depth: %.2f hops per codeword, **%.2f hops per wavefront** (a wave-uniform exit walks the largest depth of its "
                 "lanes); walks of <= 4 hops: %.1f %% of codeword steps, %.1f %% of wavefront steps."
                 % (mean_lane, mean_wave, 100.0 * lane_hist[:4].sum() / lt, 100.0 * wave_hist[:4].sum() / wt))
    # ---- all 64 survivors: after how many hops have the paths of ALL states merged into one? (what a traceback from ANY state, i.e.
    # without the first-argmin, would need) -- the survivor set as a 64-bit mask per codeword, one hop = (M & ~W, M & W) folded and
    # interleaved, W = the step's decision word
    Wd = np.zeros((n_steps + 1, B), np.uint64)
    for s_ in range(S):
        Wd |= (choice[:, :, s_].astype(np.uint64) << np.uint64(s_))

    def spread(x):
        x = x.astype(np.uint64)
        for sh, mk in ((16, 0x0000FFFF0000FFFF), (8, 0x00FF00FF00FF00FF), (4, 0x0F0F0F0F0F0F0F0F), (2, 0x3333333333333333), (1, 0x5555555555555555)):
            x = (x | (x << np.uint64(sh))) & np.uint64(mk)
        return x
    full = np.full((n_steps + 1, B), H + 1)
    m32 = np.uint64(0xFFFFFFFF)
    for t in range(H + 1, n_steps + 1):
        M = np.full(B, np.uint64(0xFFFFFFFFFFFFFFFF))
        for dd in range(1, H + 1):
            w = Wd[t - dd + 1]
            E, O = M & ~w, M & w
            M = spread((E | (E >> np.uint64(32))) & m32) | (spread((O | (O >> np.uint64(32))) & m32) << np.uint64(1))
            single = (M & (M - np.uint64(1))) == 0
            full[t] = np.where(single & (full[t] == H + 1), dd, full[t])
    fd = full[H + 1:]
    fl = np.bincount(fd.reshape(-1), minlength=H + 2)
    fw = np.bincount(fd.reshape(fd.shape[0], B // 64, 64).max(axis=2).reshape(-1), minlength=H + 2)
    lines.append("")
    lines.append("All 64 survivors (a walk that starts from ANY state instead of the first-argmin state is exact only where they have all merged "
                 "inside the walk): merged within %d hops in %.1f %% of the codeword steps (mean depth %.1f hops), i.e. NOT merged in %.1f %% of "
                 "them and in %.1f %% of the wavefront steps -- the first-argmin cannot be dropped behind a merge test either."
                 % (H, 100.0 * (1 - fl[H + 1] / fl.sum()), (fl * np.arange(H + 2)).sum() / fl.sum(), 100.0 * fl[H + 1] / fl.sum(),
                    100.0 * fw[H + 1] / fw.sum()))
    text = "\n".join(lines)
    print(text)
    if a.out:
        with open(a.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
