#!/usr/bin/env python3
"""Headline benchmark: soft-decision Viterbi, K=7 rate-1/2 (0o133, 0o171), 1024-bit blocks,
QPSK + AWGN at Eb/N0 = 3 dB, batch 65536 codewords per GPU (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the HIP Viterbi decoder over the whole per-GPU batch with the float64 LLRs
already resident in HBM.  N > 1 is weak scaling, one process per GPU, every rank decodes its own 65536-codeword batch.
The ranks come from either launcher: ``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`` (RANK /
WORLD_SIZE in the environment; WORLD_SIZE != N is an error), or plain ``python bench.py --gpus N``, which starts the N
ranks ITSELF (subprocesses, no torch) after checking that N devices are visible -- it never silently runs one GPU.
The path shards by codeword and has no exchange step, so `value` times the decode alone; the SAME K steps are then timed
again with the one collective north_star names -- an RCCL all-gather of the decoded bits (uint8, 67.5 MB per rank and
step) on the decode stream -- in every step and reported as `value_with_gather`; `comm_world` is what RCCL itself
reports for the communicator (ncclCommCount).
Collectives (closing barrier, max over ranks, error-count all-reduce, the optional all-gather) go through the
engine's own RCCL binding (``cpx_comm_*``, commpy_amd.parallel.RankComm); torch is NOT imported anywhere in this file (round 5:
the ``--comm torch`` alternative is gone -- if the RCCL communicator cannot be formed the run fails).  Rank 0 prints ONE JSON line with the contract fields plus `roofline`
(dominant kernel as reported by the library, HIP-event timed on its own stream) and `cpu_baseline`: the UNMODIFIED reference
timed on the host cores when it is present on the box ($CPX_REFERENCE_PATH, /root/reference, importable `commpy`; kind
"reference"), else the C oracle -- a line-by-line port of the same function -- on >= 4096 distinct codewords (kind "port");
the NumPy-vectorised restatement rides along as a labelled secondary.  At N = 1 the line also carries `other_configs`
(benchmarks/other_configs.py): configs[2] (turbo), one GPU's share of configs[3] (64-QAM soft demodulator -> LDPC min-sum and
sum-product) and the soft demodulator alone, each device-resident, HIP-event timed for the same K steps and checked against
the oracle after its timed region -- the headline fields are untouched by it (``--no-other-configs`` leaves it out).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MSG_BITS = 1024
EBN0_DB = 3.0
ALG_BYTES_PER_CW = 2060 * 8 + 1030 * 1      # float64 LLRs in + uint8 bits out
HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: 8.0 TB/s spec
PMC_FILES = ("r06_viterbi_c2_pmc.json", "r05_viterbi_c2_pmc.json", "r04_viterbi_c2_pmc.json")   # written by scripts/collect_pmc.py from rocprofv3 passes over this script; newest first


def synth_inputs(B, seed_msg, seed_noise):
    """random messages -> conv_encode -> QPSK (QAMModem(4), Es=2) -> AWGN at Eb/N0.
    Returns (trellis, modem, msgs [B,1024], noisy symbols [B,1030] complex128, N0)."""
    from commpy_amd.channelcoding import Trellis, conv_encode_batch
    from commpy_amd.modulation import QAMModem
    tr = Trellis(np.array([6]), np.array([[0o133, 0o171]]))
    md = QAMModem(4)
    msgs = np.random.RandomState(seed_msg).randint(0, 2, (B, MSG_BITS))
    coded = conv_encode_batch(msgs, tr)                                  # [B, 2060]
    sym = md.modulate(coded.reshape(-1)).reshape(B, -1)                  # [B, 1030]
    N0 = md.Es / (0.5 * 2 * 10 ** (EBN0_DB / 10.0))
    nrs = np.random.RandomState(seed_noise)
    noise = nrs.randn(B, sym.shape[1], 2).view(np.complex128)[..., 0]
    return tr, md, msgs, sym + np.sqrt(N0 / 2) * noise, N0


def usable_cores():
    """Cores this process may really use: affinity mask and cgroup CPU quota, not just os.cpu_count()."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def _git_head():
    """(commit, source): the commit the tree was taken from.  Where `.git` exists: `git rev-parse HEAD`, which is also written to
    `.git_head` so that the stamp travels with the tree ("git").  On the GPU box (a snapshot without `.git`) the stamp is all there
    is ("stamp": written by the post-commit hook scripts/install_hooks.sh installs, by scripts/run_gpu_round.sh before every
    gpurun call, and by this function); it names a commit whose SOURCES may differ from the snapshot's only by uncommitted edits --
    `build_id` in the same line is the digest of the sources the loaded library was really compiled from."""
    import subprocess
    stamp = os.path.join(ROOT, ".git_head")
    if os.path.exists(os.path.join(ROOT, ".git")):
        try:
            h = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True, timeout=20).stdout.strip()
            if h:
                try:
                    with open(stamp, "w") as f:
                        f.write(h + "\n")
                except OSError:
                    pass
                return h, "git"
        except Exception:
            pass
    try:
        return (open(stamp).read().strip() or None), "stamp"
    except OSError:
        return None, None


def _find_reference():
    """Directory that holds the UNMODIFIED reference package `commpy` (veeresht/CommPy), or None.  Looked for in
    $CPX_REFERENCE_PATH, then /root/reference (the build container; absent on the GPU box), then
    oracle/_ref/commpy_reference.zip -- the six hot-path files of the reference that __graft_entry__.build() packs there, byte
    for byte and with their licence, so that they travel to the GPU box with the working tree (oracle/make_ref.py; git-ignored,
    test infrastructure; a zip archive is a valid sys.path entry) --, then sys.path."""
    for cand in (os.environ.get("CPX_REFERENCE_PATH"), "/root/reference", os.path.join(ROOT, "oracle", "_ref", "commpy_reference.zip")):
        if not cand:
            continue
        if os.path.isfile(os.path.join(cand, "commpy", "channelcoding", "convcode.py")):
            return cand
        if os.path.isfile(cand) and cand.endswith(".zip"):          # the packed copy: a sys.path entry (zipimport)
            import zipfile
            try:
                with zipfile.ZipFile(cand) as z:
                    if "commpy/channelcoding/convcode.py" in z.namelist():
                        return cand
            except (OSError, zipfile.BadZipFile):
                pass
    try:
        import importlib.util
        spec = importlib.util.find_spec("commpy")
        if spec and spec.origin and os.path.isfile(os.path.join(os.path.dirname(spec.origin), "channelcoding", "convcode.py")):
            return os.path.dirname(os.path.dirname(spec.origin))
    except (ImportError, ValueError):
        pass
    return None


def _reference_worker(job):
    """One pool process: the reference's own viterbi_decode (convcode.py:661-749) on its share of the codewords; only the
    decoder calls are timed (imports, Trellis construction and IPC excluded)."""
    ref_dir, llrs = job
    os.environ.setdefault("MPLBACKEND", "Agg")
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    import warnings
    warnings.simplefilter("ignore")
    if ref_dir not in sys.path:
        sys.path.insert(0, ref_dir)
    from commpy.channelcoding import convcode as ref
    tr = ref.Trellis(np.array([6]), np.array([[0o133, 0o171]]))
    outs, t = [], 0.0
    for x in llrs:
        t0 = time.perf_counter()
        d = ref.viterbi_decode(x, tr, None, "soft")
        t += time.perf_counter() - t0
        outs.append(np.asarray(d, dtype=np.int64))
    return np.stack(outs), t


def cpu_baseline(tr, llr_sample, gpu_bits=None, budget_s=8.0):
    """The CPU path beside the GPU number, on the host cores of THIS machine, bounded samples of the same workload:

    * kind "reference": the unmodified Python reference (convcode.py:661-749), one process per usable core
      (multiprocessing, spawn), 8 codewords per core, when the package is present on the box ($CPX_REFERENCE_PATH,
      /root/reference or an importable `commpy`); its output is compared with the engine's on those codewords;
    * kind "port" otherwise (the GPU box has no copy of the reference): oracle/cpx_oracle.c, a line-by-line C restatement of
      the same function, one thread per core over >= 4096 DISTINCT codewords (nothing is cache-hot by construction);
    * `secondary`: the batch-vectorised NumPy restatement (oracle/np_viterbi.py), single thread -- labelled, never the baseline.
    """
    import concurrent.futures as cf
    import oracle
    from oracle import np_viterbi
    oracle.load()
    cores = usable_cores()
    res = None
    ref_dir = _find_reference()
    if ref_dir:
        import multiprocessing as mp
        per = 8
        n = min(len(llr_sample), cores * per)
        jobs = [(ref_dir, llr_sample[i:n:cores]) for i in range(cores) if i < n]
        t0 = time.perf_counter()
        with mp.get_context("spawn").Pool(len(jobs)) as pool:
            parts = pool.map(_reference_worker, jobs)
        wall = time.perf_counter() - t0
        busy = sum(t for _, t in parts)
        slowest = max(t for _, t in parts)
        mism = None
        if gpu_bits is not None:
            mism = int(sum(np.sum(out != gpu_bits[i:n:cores][:len(out)]) for i, (out, _) in enumerate(parts)))
        res = {"value": n * MSG_BITS / slowest, "unit": "info-bits/s", "cores": len(jobs), "kind": "reference",
               "per_core": n * MSG_BITS / busy,
               "sample": "%d codewords (K=7 soft, 1024-bit; the first of the timed batch) through the unmodified "
                         "commpy.channelcoding.viterbi_decode from %s, %d processes x %d codewords; %.1f s of decoder calls "
                         "in the slowest process (%.1f s wall incl. process start-up and imports, not counted); "
                         "bits differing from the engine's on these codewords: %s"
                         % (n, ref_dir, len(jobs), per, slowest, wall, mism)}
    # the C port: distinct codewords only
    nd = min(len(llr_sample), 4096)
    chunks = [np.ascontiguousarray(llr_sample[i:nd:cores]) for i in range(cores)]
    chunks = [c for c in chunks if len(c)]
    deadline = [0.0]

    def work(c):
        done, pos, step = 0, 0, 8
        while time.perf_counter() < deadline[0]:                  # time-bounded: every call is ONE C call (GIL released)
            blk = c[pos:pos + step]
            if not len(blk):
                pos = 0                                           # (only a box much faster than expected wraps around)
                continue
            oracle.viterbi_decode(blk, tr, None, "soft")
            done += len(blk)
            pos += step
        return done

    with cf.ThreadPoolExecutor(len(chunks)) as ex:
        t0 = time.perf_counter()
        deadline[0] = t0 + budget_s
        n = sum(ex.map(work, chunks))
        dt = time.perf_counter() - t0
    port = {"value": n * MSG_BITS / dt, "unit": "info-bits/s", "cores": len(chunks), "kind": "port",
            "sample": "%d codeword decodes (K=7 soft, 1024-bit) over %d distinct codewords of the timed batch through "
                      "oracle/cpx_oracle.c orc_viterbi_decode on %d threads, %.1f s wall" % (n, nd, len(chunks), dt)}
    if res is None:
        res = port
        res["sample"] += ("; the unmodified Python reference is not present on this box (set CPX_REFERENCE_PATH to time "
                          "it) -- measured in the build container it does ~1.0e3 info-bits/s per core (BASELINE.md)")
    else:
        res["port"] = port
    nb = min(len(llr_sample), 256)
    t0 = time.perf_counter()
    got = np_viterbi.viterbi_decode_batch(llr_sample[:nb], tr, None, "soft")
    dt = time.perf_counter() - t0
    res["secondary"] = {"value": nb * MSG_BITS / dt, "unit": "info-bits/s", "cores": 1,
                        "kind": "numpy-vectorised restatement (oracle/np_viterbi.py), NOT the reference",
                        "sample": "%d codewords in one batched call, %.2f s; equal to the C port: %s"
                                  % (nb, dt, bool(np.array_equal(got, oracle.viterbi_decode(llr_sample[:nb], tr, None, "soft"))))}
    return res


def free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def launch_ranks(n, child_argv, env=None, timeout=None):
    """Start the `n` ranks of a one-node job (one process per GPU) and wait for them: what `torch.distributed.run` does for
    this script, without torch.  Every child gets RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR (127.0.0.1) /
    MASTER_PORT and one CPX_COMM_NONCE per launch (the RCCL id exchange keys on it, commpy_amd/parallel.py).  stdout and
    stderr are inherited: rank 0 prints the JSON line.  When a rank fails, the others -- who would wait for it in the next
    collective -- are terminated (by pid) and the first non-zero exit code is returned."""
    import subprocess
    import uuid
    base = dict(os.environ if env is None else env)
    base.update({"WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                 "MASTER_PORT": str(free_port()), "CPX_COMM_NONCE": uuid.uuid4().hex})
    procs = []
    for r in range(n):
        e = dict(base)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r)})
        procs.append(subprocess.Popen(list(child_argv), env=e))
    deadline = None if timeout is None else time.time() + timeout
    rc = 0
    live = list(procs)
    while live:
        for pr in list(live):
            code = pr.poll()
            if code is None:
                continue
            live.remove(pr)
            if code != 0 and rc == 0:
                rc = code
        if live and (rc != 0 or (deadline is not None and time.time() > deadline)):
            if rc == 0:
                rc = 124
            for pr in live:
                pr.terminate()
            for pr in live:
                try:
                    pr.wait(10)
                except subprocess.TimeoutExpired:
                    pr.kill()
                    pr.wait()
            live = []
        if live:
            time.sleep(0.05)
    return rc


def visible_devices():
    """Number of HIP devices the engine sees (0 when there is none or the library cannot initialise one)."""
    from commpy_amd import _lib
    try:
        return int(_lib.device_count())
    except Exception:
        return 0


def resolve_world(args, environ=None, n_devices=None):
    """Which of the three ways this invocation runs: ("single", 0, 1), ("rank", rank, world) or ("launch", 0, N).
    Errors (SystemExit with a message) instead of EVER running fewer GPUs than --gpus asks for."""
    env = os.environ if environ is None else environ
    if "RANK" in env and "WORLD_SIZE" in env:
        rank, world = int(env["RANK"]), int(env["WORLD_SIZE"])
        if world != args.gpus:
            raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks (rank %d): they must agree"
                             % (args.gpus, world, rank))
        return "rank", rank, world
    if args.gpus <= 1:
        return "single", 0, 1
    have = visible_devices() if n_devices is None else n_devices
    if have < args.gpus:
        raise SystemExit("bench.py: --gpus %d needs %d visible MI355X devices, this box shows %d -- refusing to run a "
                         "smaller job under that name" % (args.gpus, args.gpus, have))
    return "launch", 0, args.gpus


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=65536, help="codewords per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-path-check", action="store_true",
                    help="skip the untimed full-batch comparison of the three Viterbi kernel families before the warm-up")
    ap.add_argument("--gather", action="store_true",
                    help="accepted for older command lines; N > 1 always times both regions (decode alone -> value, decode + "
                         "RCCL all-gather of the decoded bits on the decode stream -> value_with_gather)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the second (decode + all-gather) timed region")
    ap.add_argument("--sustain-seconds", type=float, default=2.0,
                    help="N = 1: after the timed region run the same step back to back for this long (GPU-busy evidence; 0 = skip)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N = 1: skip the `other_configs` lines (turbo, LDPC chain, soft demodulator; ~15 s)")
    ap.add_argument("--synth", choices=("device", "host"), default="device",
                    help="where the synthetic input is generated: on the GPU (Philox bits/noise, device encoder and "
                         "modulator; fast, no large host arrays) or on the host (NumPy MT19937, SURVEY 8d seeds)")
    ap.add_argument("--precision", choices=("fp64-parity", "fp32-fast"), default="fp64-parity",
                    help="fp32-fast: the float32 variant of the fused kernel (cpx_set_precision) -- NOT the parity mode and not the "
                         "headline: the line then carries dtype f32 and the measured mismatch count against the float64 oracle")
    args = ap.parse_args()

    mode, rank, world = resolve_world(args)
    if mode == "launch":
        # plain `python bench.py --gpus N`: this process only starts the N ranks (same script, same arguments) and waits
        sys.exit(launch_ranks(world, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    distributed = mode == "rank"
    args.gather = distributed and world > 1 and not args.no_gather
    comm = None
    from commpy_amd import _lib
    lib = _lib.load()
    _lib.require_device()
    _lib.check(lib.cpx_set_device(local_rank))
    _lib.set_precision(args.precision)
    if distributed:
        # the engine's own RCCL binding (the only collective path: north_star allows no PyTorch); if the communicator cannot
        # be formed the run FAILS
        from commpy_amd.parallel import RankComm
        comm = RankComm(rank, world)

    B = args.batch
    nsym = (MSG_BITS + 6) * 2 // 2                                 # 1030 QPSK symbols per codeword
    LEN, L, T, TB = 2 * nsym, nsym, nsym + 6 - 1, 30            # 2060 LLRs -> 1030 bits, 1035 steps, tb = min(5m, L)
    if args.synth == "host":
        tr, md, msgs, y, N0 = synth_inputs(B, 10 + 1000 * rank, 11 + 1000 * rank)
        y = np.ascontiguousarray(y)
    else:
        from commpy_amd.channelcoding import Trellis
        from commpy_amd.modulation import QAMModem
        tr = Trellis(np.array([6]), np.array([[0o133, 0o171]]))
        md = QAMModem(4)
        N0 = md.Es / (0.5 * 2 * 10 ** (EBN0_DB / 10.0))
        msgs = y = None
    h_tr, h_md = tr._device_handle(), md._device_handle()

    d_full = ctypes.c_void_p()                                      # --gather: [world][B][L] uint8, every rank's bits
    stream = None                                                   # the library's own stream (cpx_default_stream)
    d_y, d_llr = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.check(lib.cpx_malloc(ctypes.byref(d_y), B * nsym * 16))
    _lib.check(lib.cpx_malloc(ctypes.byref(d_llr), B * LEN * 8))
    _lib.check(lib.cpx_malloc(ctypes.byref(d_full), (world if args.gather else 1) * B * L))
    d_bits = ctypes.c_void_p(d_full.value + (rank * B * L if args.gather else 0))   # this rank's slot (in-place gather)

    def sync():
        _lib.check(lib.cpx_stream_sync(None))

    def barrier():
        if comm is not None:
            comm.barrier()                                          # RCCL all-reduce of one word + stream sync

    if args.synth == "host":
        _lib.check(lib.cpx_memcpy_h2d(d_y, _lib.ptr(y), y.nbytes))
    else:
        # random bits -> conv_encode -> QPSK -> AWGN, all on the device (csrc/linksim.hip)
        d_msg, d_coded, d_sym = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(lib.cpx_malloc(ctypes.byref(d_msg), B * MSG_BITS))
        _lib.check(lib.cpx_malloc(ctypes.byref(d_coded), B * LEN))
        _lib.check(lib.cpx_malloc(ctypes.byref(d_sym), B * nsym * 16))
        _lib.check(lib.cpx_random_bits_dev(d_msg, B * MSG_BITS, 10 + 1000 * rank, 1, stream))
        _lib.check(lib.cpx_conv_encode_batch_dev(h_tr, d_msg, B, MSG_BITS, 1, 0, d_coded, LEN, stream))
        _lib.check(lib.cpx_modulate_dev(h_md, d_coded, B * nsym, d_sym, stream))
        sc = float(np.sqrt(N0 / 2))
        _lib.check(lib.cpx_awgn_dev(d_sym, B * nsym, sc, sc, 11 + 1000 * rank, 2, d_y, stream))
        sync()
        msgs = np.empty((B, MSG_BITS), dtype=np.uint8)
        _lib.check(lib.cpx_memcpy_d2h(_lib.ptr(msgs), d_msg, msgs.nbytes))
        y = np.empty((1, nsym), dtype=np.complex128)              # first codeword's symbols for the demod spot check
        _lib.check(lib.cpx_memcpy_d2h(_lib.ptr(y), d_y, y.nbytes))
        for d in (d_msg, d_coded, d_sym):
            _lib.check(lib.cpx_free(d))

    # LLRs are produced on the device by the soft demodulator (same formula as Modem.demodulate) and stay in HBM.
    _lib.check(lib.cpx_demod_soft_dev(h_md, d_y, B * nsym, float(N0), d_llr, stream))
    sync()

    # one HIP-event pair per timed step, recorded on the launch stream and read AFTER the timed region
    def make_timers(n):
        out = []
        for _ in range(n):
            tmr = ctypes.c_void_p()
            _lib.check(lib.cpx_timer_create(ctypes.byref(tmr)))
            out.append(tmr)
        return out

    def read_timers(tmrs):
        out = []
        for tmr in tmrs:
            ms = ctypes.c_float()
            _lib.check(lib.cpx_timer_elapsed_ms(tmr, ctypes.byref(ms)))
            out.append(ms.value)
        return out

    # Untimed, before the warm-up: the three Viterbi kernel families on the WHOLE batch that is timed below -- the default dispatch
    # (the fused kernel), the two-kernel codeword path and the state-per-lane kernels must return the same bits for all B codewords
    # (the oracle comparison after the timed region covers 16 386 of them).  Said plainly: these ~7 ms of decoding also mean that the
    # GPU is no longer at its idle clock when the W warm-up steps start (BENCH_DEBUG=1 prints the per-step times; --no-path-check
    # leaves the check out).
    path_check = None
    if not args.no_path_check and args.precision == "fp64-parity":
        d_alt, d_errs = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(lib.cpx_malloc(ctypes.byref(d_alt), B * L))
        _lib.check(lib.cpx_malloc(ctypes.byref(d_errs), B * 4))
        _lib.check(lib.cpx_viterbi_decode_batch_dev(h_tr, d_llr, B, LEN, L, T, TB, 1, d_bits, stream))
        names, mism = [_lib.viterbi_last_path()], {}
        errs = np.empty(B, dtype=np.int32)
        for path in ("cw2", "wave"):
            _lib.viterbi_set_path(path)
            try:
                _lib.check(lib.cpx_viterbi_decode_batch_dev(h_tr, d_llr, B, LEN, L, T, TB, 1, d_alt, stream))
                names.append(_lib.viterbi_last_path())
            finally:
                _lib.viterbi_set_path(None)
            _lib.check(lib.cpx_count_errors_dev(d_bits, L, d_alt, L, B, 1, L, d_errs, stream))
            sync()
            _lib.check(lib.cpx_memcpy_d2h(_lib.ptr(errs), d_errs, errs.nbytes))
            mism[path] = int(errs.sum())
        # ... and the default dispatch against ITSELF: REPEAT more launches over the same input must reproduce the first launch's bits
        # exactly (a race between the traceback slotted into step t and the ring writes of step t + 1 would show up here and nowhere
        # else).  Untimed like the rest of this block; said plainly again: together with the family comparison above this is ~25 ms of
        # decoding in front of the W warm-up steps, i.e. most of a fresh process's clock ramp happens here and not in the timed region.
        REPEAT = 12
        rep_mism = 0
        for _ in range(REPEAT):
            _lib.check(lib.cpx_viterbi_decode_batch_dev(h_tr, d_llr, B, LEN, L, T, TB, 1, d_alt, stream))
            _lib.check(lib.cpx_count_errors_dev(d_bits, L, d_alt, L, B, 1, L, d_errs, stream))
            sync()
            _lib.check(lib.cpx_memcpy_d2h(_lib.ptr(errs), d_errs, errs.nbytes))
            rep_mism += int(errs.sum())
        path_check = {"codewords": B, "kernel_paths": names, "mismatching_bits_vs_default_dispatch": mism,
                      "repeat_launches": REPEAT, "mismatching_bits_between_repeats": rep_mism}
        for d in (d_alt, d_errs):
            _lib.check(lib.cpx_free(d))

    def step(tmr, gather):
        if tmr is not None:
            _lib.check(lib.cpx_timer_start(tmr, stream))
        _lib.check(lib.cpx_viterbi_decode_batch_dev(h_tr, d_llr, B, LEN, L, T, TB, 1, d_bits, stream))
        if gather:                                                  # the collective north_star names, on the decode stream
            comm.allgather_dev(d_bits, d_full, B * L, stream)
        if tmr is not None:
            _lib.check(lib.cpx_timer_stop(tmr, stream))

    def timed_region(gather):
        """W warm-up steps, (N > 1: an untimed rehearsal of the whole region,) then EXACTLY K steps between barrier +
        synchronize on both sides; returns (elapsed seconds, MAX over ranks; per-step event times of this rank)."""
        warm = make_timers(1)[0]
        # warm-up runs exactly what a timed step runs, event records included
        for _ in range(max(args.warmup, 1)):
            step(warm, gather)
        barrier(); sync()
        if distributed:
            # untimed rehearsal of the whole timed region (same K launches, same closing barrier + synchronize): one-time
            # host-side costs of the collective path (seen sporadically as a ~70 ms stall in the first closing barrier of
            # a process on a fresh box) land here, not in the measurement
            for k in range(args.steps):
                step(warm, gather)
            barrier(); sync()
        tmrs = make_timers(args.steps)
        barrier(); sync()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(tmrs[k], gather)
        t1 = time.perf_counter()
        barrier()
        t2 = time.perf_counter()
        sync()
        el = time.perf_counter() - t0
        if os.environ.get("BENCH_DEBUG"):
            print("debug[gather=%s]: enqueue %.3f ms, barrier %.3f ms, sync %.3f ms" % (
                gather, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t0 + el - t2) * 1e3), file=sys.stderr)
        ms = read_timers(tmrs)
        if os.environ.get("BENCH_DEBUG"):
            print("debug[gather=%s]: per-step launch ms " % gather + " ".join("%.3f" % v for v in ms), file=sys.stderr)
        if comm is not None:
            el = float(comm.allreduce(np.array([el]), "max")[0])                    # MAX over ranks
        for tmr in tmrs + [warm]:
            lib.cpx_timer_destroy(tmr)
        return el, ms

    elapsed, kernel_ms = timed_region(False)                       # `value`: the decode path alone (no data-path collective)
    kernel_name = _lib.last_kernel()                               # what the library really launched for this workload
    elapsed_g = gather_ms = None
    if args.gather:
        elapsed_g, gather_ms = timed_region(True)                  # `value_with_gather`: the same K steps + the all-gather
    # Clock evidence (round 5), untimed: the same K steps once more with the engine's shader-clock probe (one sleeping wavefront on its own
    # stream, cpx_sclk_probe_*) running alongside: average sclk over the interval and the per-launch times measured DURING it.  sclk x
    # time = cycles per launch, which is what rocprofv3's GRBM_GUI_ACTIVE / 8 reports for the same kernel at the lower clock it runs
    # at under the profiler (profiles/README.md) -- recorded instead of argued.
    clock = None
    probe = ctypes.c_void_p()
    try:
        probe = ctypes.c_void_p()                                   # a throw-away probe first: its one-time costs (stream, buffer, code
        _lib.check(lib.cpx_sclk_probe_start(ctypes.byref(probe), 0.2))   # object) must not leave the GPU idle in front of the real interval
        _lib.check(lib.cpx_sclk_probe_read(probe, None, None))
        probe = ctypes.c_void_p()
        for _ in range(max(args.warmup, 8)):                       # the host read the timers meanwhile: no idle clock in the interval
            step(None, False)
        sync()
        probe = ctypes.c_void_p()
        _lib.check(lib.cpx_sclk_probe_start(ctypes.byref(probe), max(0.5, 0.9 * args.steps * float(np.median(kernel_ms)))))
        tm2 = make_timers(args.steps)
        for k in range(args.steps):
            step(tm2[k], False)
        sync()
        ms2 = read_timers(tm2)
        for tmr in tm2:
            lib.cpx_timer_destroy(tmr)
        mhz, ival = ctypes.c_double(), ctypes.c_double()
        started, probe = probe, ctypes.c_void_p()                   # (read frees the probe whatever it returns)
        _lib.check(lib.cpx_sclk_probe_read(started, ctypes.byref(mhz), ctypes.byref(ival)))
        clock = {"sclk_mhz": mhz.value, "probe_interval_ms": ival.value, "kernel_ms_avg_during_probe": float(np.mean(ms2)),
                 "kernel_ms_median_during_probe": float(np.median(ms2)),
                 "shader_cycles_per_launch": mhz.value * 1e3 * float(np.median(ms2)),
                 "how": "cpx_sclk_probe: s_memtime (shader-clock counter) over s_memrealtime (constant-rate counter) of one sleeping "
                        "wavefront on its own stream while the same K steps ran again, outside the timed region"}
    except Exception as exc:                                       # evidence, never a reason to lose the line
        clock = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
        if probe:                                                  # a probe that was started and never read goes back to the library
            lib.cpx_sclk_probe_destroy(probe)
    # Sustained phase (round 6), untimed for `value`: the K timed steps are ~30 ms of a run that lasts many seconds, so an outside sampler
    # (rocm-smi style GPU-busy polling, the driver's own clock) cannot see them.  The same step then runs back to back for
    # --sustain-seconds of wall time; its average per step is reported next to the timed one and has to agree with it.
    sustained = None
    if world == 1 and args.sustain_seconds > 0:
        try:
            t0s, n_s = time.perf_counter(), 0
            while time.perf_counter() - t0s < args.sustain_seconds:
                for _ in range(100):
                    step(None, False)
                sync()
                n_s += 100
            dts = time.perf_counter() - t0s
            sustained = {"steps": n_s, "seconds": dts, "ms_per_step": dts / n_s * 1e3, "info_bits_per_s": B * MSG_BITS * n_s / dts,
                         "note": "the headline step back to back (host clock around batches of 100 launches + a stream sync), after the timed "
                                 "region; not part of `value`"}
        except Exception as exc:                                   # evidence, never a reason to lose the line
            sustained = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    comm_world = None
    if comm is not None:
        nr, nl, fr = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(lib.cpx_comm_info(comm.h, ctypes.byref(nr), ctypes.byref(nl), ctypes.byref(fr)))
        comm_world = {"nranks": nr.value, "this_rank": fr.value, "source": "ncclCommCount / ncclCommUserRank"}

    # ---- correctness of what was timed: BER vs the messages, parity vs the oracle on a sample ----
    bits = np.empty((B, L), dtype=np.uint8)
    _lib.check(lib.cpx_memcpy_d2h(_lib.ptr(bits), d_bits, bits.nbytes))
    nerr = int(np.sum(bits[:, :MSG_BITS] != msgs))
    gather_ok = None

    def allreduce_i64(v):
        v = np.ascontiguousarray(v, dtype=np.int64)
        return comm.allreduce(v)

    if world > 1:
        # error counts of all shards: an RCCL all-reduce of int64 counters (links.py:252-260), outside the timed region
        nerr = int(allreduce_i64([nerr])[0])
    if args.gather:
        # every rank holds every rank's bits: this rank's own slot must equal what it decoded, and every OTHER slot must carry
        # the checksums (ones count, position-weighted sum) its owner computed from its own decode and published by all-reduce
        full = np.empty((world, B, L), dtype=np.uint8)
        _lib.check(lib.cpx_memcpy_d2h(_lib.ptr(full), d_full, full.nbytes))
        w = (np.arange(L, dtype=np.int64) % 251) + 1

        def cks(a):
            return [int(a.sum(dtype=np.int64)), int((a.sum(axis=0, dtype=np.int64) * w).sum())]

        mine = np.zeros((world, 2), np.int64)
        mine[rank] = cks(bits)
        published = allreduce_i64(mine.reshape(-1)).reshape(world, 2)
        gather_ok = bool(np.array_equal(full[rank], bits) and
                         all(cks(full[r]) == list(published[r]) for r in range(world)))
        gather_ok = bool(allreduce_i64([0 if gather_ok else 1])[0] == 0)         # true only if it held on EVERY rank
    ber = nerr / float(world * B * MSG_BITS)
    out = None
    if rank == 0:
        import oracle
        # parity of what was timed: 16384 codewords (first / middle / last 5462 of the batch: every wavefront position of
        # the first, middle and last workgroups) against the oracle, decoded on the host threads
        ns = min(B, 5462)
        starts = sorted({0, max(0, B // 2 - ns // 2), B - ns})
        mism, checked = 0, 0
        llr_s = None
        for lo in starts:
            blk = np.empty((ns, LEN))
            _lib.check(lib.cpx_memcpy_d2h(_lib.ptr(blk), ctypes.c_void_p(d_llr.value + lo * LEN * 8), blk.nbytes))
            want = oracle.viterbi_decode_mt(blk, tr, None, "soft", usable_cores())
            mism += int(np.sum(want != bits[lo:lo + ns]))
            checked += ns
            if lo == 0:
                llr_s = blk
        llr_ref = oracle.demodulate(md.constellation, y[0], "soft", N0)
        demod_err = float(np.max(np.abs(llr_ref - llr_s[0])))
        kavg = float(np.mean(kernel_ms))
        achieved = ALG_BYTES_PER_CW * B / (kavg * 1e-3) / 1e9
        value = world * B * MSG_BITS * args.steps / elapsed
        value_g = None if elapsed_g is None else world * B * MSG_BITS * args.steps / elapsed_g
        # HBM traffic and VALU occupancy of this kernel from the committed rocprofv3 PMC passes (scripts/collect_pmc.py).  They
        # cannot be measured inside this run (counters need the profiler), so they are quoted ONLY when the file was recorded
        # by a library built from the same Viterbi sources as the one loaded now (cpx_build_id, "viterbi" digest), for the same
        # kernel and batch; otherwise null.
        traffic = traffic_src = valu = None
        build = _lib.build_id()
        for pmc_file in PMC_FILES:
            try:
                with open(os.path.join(ROOT, "profiles", pmc_file)) as f:
                    tj = json.load(f)
                same = (tj.get("batch") == B and kernel_name.split("<")[0] in tj.get("kernel", "?") and
                        args.precision == "fp64-parity" and                  # (the counters were taken in parity mode)   rocprofv3 prints the
                        # template arguments as numbers ("<6, 109u, 79u, 1, 28>"): the kernel's base name and the batch identify it
                        bool(build.get("viterbi")) and (tj.get("build_id") or {}).get("viterbi") == build.get("viterbi"))
                if same:
                    traffic = tj["traffic_bytes_per_launch"]
                    traffic_src = ("profiles/%s (rocprofv3 FETCH_SIZE / WRITE_SIZE passes; gfx950 correction as in the file; "
                                   "recorded at git %s with the same Viterbi sources, build id %s)"
                                   % (pmc_file, str(tj.get("git_head"))[:12], build.get("viterbi")))
                    valu = tj.get("valu")
                    break
            except (OSError, ValueError, KeyError):
                pass
        out = {
            "metric": "decoded info-bits/s at fixed Eb/N0 (Viterbi K=7 r=1/2, 1024b); BER match",
            "value": value, "unit": "info-bits/s", "n_gpus": world if distributed else 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f64" if args.precision == "fp64-parity" else "f32 (fp32-fast mode: not bit-exact, see oracle_mismatched_bits)",
            "data": "synthetic (%s-generated: random messages -> conv_encode -> QPSK -> AWGN -> soft demod on device)" % args.synth,
            "config": {"workload": "configs[1]: K=7 (0o133,0o171) r=1/2, 1024-bit blocks, soft Viterbi over "
                                   "AWGN+QPSK at Eb/N0=3 dB, batch=%d codewords per GPU, tb_depth=30" % B,
                       "batch_per_gpu": B, "block_bits": MSG_BITS, "ebn0_db": EBN0_DB,
                       "parallelism": "codewords sharded x%d, no data-path collective in `value`%s" % (
                           world, "; `value_with_gather`: + one RCCL all-gather of the decoded bits (%d B per rank) per step"
                           % (B * L) if args.gather else ""),
                       "collectives": "engine RCCL binding (cpx_comm_*)" if comm is not None else "none (single process)"},
            "ber": ber, "oracle_mismatched_bits": mism, "oracle_sample_codewords": checked, "kernel_path_check": path_check, "build_id": build,
            "git_head": _git_head()[0], "git_head_source": _git_head()[1],
            "demod_max_abs_err_vs_oracle": demod_err,
            # the kernel is bound by VALU issue, not by HBM: achieved / peak / frac are the HBM figures the
            # contract asks for, `valu` carries the ceiling that actually binds (from the PMC passes in profiles/)
            "roofline": {"bound": "valu", "kernel": kernel_name, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel_ms_avg": kavg, "kernel_ms_min": float(np.min(kernel_ms)),
                         "kernel_ms_median": float(np.median(kernel_ms)),
                         "kernel_ms_scope": "HIP events around one decode call on its stream: the decoder kernel plus the "
                                            "NaN-redo launch that follows it (~4 us when nothing is flagged)",
                         # `value` above is the contract's number (K steps after W warm-ups).  In a fresh process the first dozen
                         # launches run while the GPU is still raising its clock (per-launch times fall from ~1.78 to ~1.53 ms
                         # over 20 ms, BENCH_DEBUG=1 prints them), so short runs include part of that ramp; the median launch is
                         # what a long-running decoder sees.
                         "steady_state": {"ms_per_launch_median": float(np.median(kernel_ms)),
                                          "info_bits_per_s_per_gpu": B * MSG_BITS / (float(np.median(kernel_ms)) * 1e-3)},
                         "algorithmic_bytes_per_launch": ALG_BYTES_PER_CW * B,
                         "clock": clock, "sustained": sustained,
                         "valu": valu,
                         "note": "serial float64 add-compare-select recursion: VALU-issue bound (valu.busy_frac), the HBM "
                                 "fraction is reported because the metric asks for it"},
        }
        if distributed:
            # N > 1: the same K steps with north_star's one collective (all-gather of the decoded bits) on the decode stream in
            # every step; comm_world = what the communicator itself reports, so "did RCCL see N ranks" is answerable from here
            out["value_with_gather"] = value_g
            out["ms_per_step_with_gather"] = None if elapsed_g is None else elapsed_g / args.steps * 1e3
            out["gather"] = None if gather_ms is None else {
                "bytes_per_rank_per_step": B * L, "step_ms_avg_rank0": float(np.mean(gather_ms)),
                "all_slots_ok_on_all_ranks": gather_ok,
                "note": "HIP events on rank 0 around decode + ncclAllGather (in place, [world][B][L] uint8 on every GPU)"}
            out["comm_world"] = comm_world
            out["launcher"] = "bench.py (subprocess per rank)" if os.environ.get("CPX_COMM_NONCE") else "external (RANK/WORLD_SIZE set)"
        # the other decoders north_star names, on their own configurations (N = 1 only; the headline fields above are final)
        out["other_configs"] = None
        if world == 1 and not args.no_other_configs and args.precision == "fp64-parity":
            for d in (d_y, d_llr, d_full):
                _lib.check(lib.cpx_free(d))
            d_y = d_llr = d_full = None
            from benchmarks import other_configs
            out["other_configs"] = other_configs.run(lib, args.steps, args.warmup,
                                                     log=(lambda m: print(m, file=sys.stderr)) if os.environ.get("BENCH_DEBUG") else None)
        if not args.no_cpu_baseline and world == 1:                # reported at N = 1 only; the other ranks would idle
            out["cpu_baseline"] = cpu_baseline(tr, llr_s, bits[:ns, :].astype(np.int64))
        else:
            out["cpu_baseline"] = None
    if comm is not None:
        comm.barrier()
        comm.close()
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
