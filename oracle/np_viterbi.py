"""Batch-vectorised NumPy restatement of the reference's Viterbi decoder -- TEST INFRASTRUCTURE ONLY.

The second CPU baseline of bench.py ("our vectorised NumPy restatement, clearly labelled") and a
cross-check of the C oracle.  Only tests/ and bench.py's cpu_baseline leg may import it; the product never does.

It follows the decision rule of /root/reference/commpy/channelcoding/convcode.py:661-749 in the collapsed form
SURVEY Appendix A.1 verified against the live reference:

    forward:   cand_j[s] = pm[pred_j(s)] + bm(output of that branch);  first minimum wins (np.where order, :561-572,
               :633-642);  best[t] = first argmin over the states (:645)
    output:    the k bits of step s come from the survivor at step s of the path traced back from
               best[min(s + tb_depth - 2, t_max)]  (the reference re-runs a full traceback every step and later
               tracebacks overwrite earlier ones, :644-657)

vectorised over the batch axis (the reference decodes one codeword per call): the time loop stays a Python loop, every
operation inside it is one NumPy call over all B codewords.  Branch metrics are the reference's formulas evaluated in
its order (:575-587: hamming / log(exp(r) + 1) / squared distance, summed MSB first from 0).
"""
import numpy as np


def _pred_tables(trellis):
    """Predecessors of every state in np.where (row-major) order: [S, I] previous state, input, output codeword."""
    nxt = np.asarray(trellis.next_state_table)
    out = np.asarray(trellis.output_table)
    S, I = nxt.shape
    ps = np.zeros((S, I), np.int64)
    pi = np.zeros((S, I), np.int64)
    pc = np.zeros((S, I), np.int64)
    cnt = np.zeros(S, np.int64)
    for p in range(S):
        for i in range(I):
            s = nxt[p, i]
            ps[s, cnt[s]], pi[s, cnt[s]], pc[s, cnt[s]] = p, i, out[p, i]
            cnt[s] += 1
    assert np.all(cnt == I), "every state needs number_inputs incoming branches (convcode.py:604-629)"
    return ps, pi, pc


def viterbi_decode_batch(coded, trellis, tb_depth=None, decoding_type='hard'):
    """coded: [B, len] float64 -> int64 [B, L]; same arguments as convcode.py:661."""
    x = np.ascontiguousarray(coded, dtype=np.float64)
    B, length = x.shape
    k, n, m = int(trellis.k), int(trellis.n), int(trellis.total_memory)
    S = int(trellis.number_states)
    L = int(length * (k / n))                                        # (:698)
    tb = int(tb_depth) if tb_depth else min(5 * m, L)                # (:701-702)
    t_max = int((L + m) / k) - 1                                     # range(1, int((L + m) / k)) (:721)
    if decoding_type == 'soft':
        with np.errstate(invalid='ignore'):
            x = np.clip(x, -500, 500)                                # (:718-719); NaN passes
    ps, pi, pc = _pred_tables(trellis)
    cw_bits = (np.arange(1 << n)[:, None] >> (n - 1 - np.arange(n))[None, :]) & 1    # dec2bitarray(c, n), MSB first (:622)
    pad = -1.0 if decoding_type == 'unquantized' else 0.0            # (:726-734)
    pm = np.full((B, S), np.inf)
    pm[:, 0] = 0.0                                                   # (:705-706)
    choice = np.zeros((t_max + 1, B, S), np.int8)                    # index of the winning predecessor
    best = np.zeros((t_max + 1, B), np.int64)
    rows = np.arange(B)[:, None]
    with np.errstate(over='ignore', invalid='ignore'):
        for t in range(1, t_max + 1):
            r = x[:, (t - 1) * n:t * n] if t <= L // k else np.full((B, n), pad)     # (:722-734)
            if decoding_type == 'hard':
                per = (r.astype(np.int64)[:, None, :] ^ cw_bits[None, :, :]).astype(np.float64)      # hamming_dist (:580)
            elif decoding_type == 'soft':
                nll0 = np.log(np.exp(r) + 1)                         # (:582)
                nll1 = nll0 - r                                      # (:583)
                per = np.where(cw_bits[None, :, :] == 1, nll1[:, None, :], nll0[:, None, :])          # (:584)
            else:
                d = r[:, None, :] - (2.0 * cw_bits[None, :, :] - 1.0)                                # (:586-587)
                per = d * d
            bm = np.zeros((B, 1 << n))
            for j in range(n):                                       # n < 8: NumPy's add.reduce is sequential from 0
                bm = bm + per[:, :, j]
            cand = pm[:, ps] + bm[rows, pc.reshape(-1)].reshape(B, S, -1)            # (:629)
            ch = np.argmin(cand, axis=2)                             # first minimum; all-NaN -> 0 (:633-642)
            pm = np.take_along_axis(cand, ch[:, :, None], axis=2)[:, :, 0]
            choice[t] = ch
            best[t] = np.argmin(pm, axis=1)                          # (:645)
    out = np.zeros((B, max(t_max * k, L)), np.int64)
    brange = np.arange(B)
    H = tb - 2
    # the walks from best[t0] for output steps s with the same number of hops advance together
    for s in range(1, t_max + 1):
        t0 = min(s + H, t_max)
        st = best[t0]
        for tt in range(t0, s, -1):
            st = ps[st, choice[tt, brange, st]]
        sym = pi[st, choice[s, brange, st]]                          # decoded_symbols (:650)
        for b in range(k):
            out[:, (s - 1) * k + b] = (sym >> (k - 1 - b)) & 1       # dec2bitarray(sym, k) (:652)
    return out[:, :L]
