"""Multi-GPU build+probe: shard by key radix, one all-to-all per relation, local tables.

New design (the reference is single-process shared memory). One process per GPU
(torch.distributed; backend "nccl" is RCCL over xGMI on ROCm). Rank g holds the g-th contiguous
piece of R and of S. Per join:

  1. histogram   destination of a tuple = one radix digit of its key (HASH_BIT_MODULO, mc/src/parallel_radix_join.c:59):
                 the low log2 G bits, or the high ones = a range split (see ShardedJoin)  -> hj_shard_histogram_dev
  2. counts      one all_to_all_single of the G per-destination counts (R and S together)
  3. scatter     tuples grouped by destination IN INPUT ORDER, written as bare 32-bit keys (a DataGen
                 tuple is its key): 4 bytes per tuple cross the links            -> hj_shard_scatter_dev
  4. exchange    one all-to-all per relation, issued as a batch of direct pairwise sends/receives of at
                 most 512 MiB (async: R's exchange overlaps the split of S, S's exchange overlaps the
                 local build); every rank sends 1/G of its tuples to every peer directly, so all xGMI
                 links carry traffic at once (no ring). The receiver lays the pieces out in source-rank
                 order: since rank g holds the g-th piece of the relation and the scatter keeps input
                 order, position in the received buffer is GLOBAL input order -- the reference's
                 insertion order survives the exchange without any index travelling
  5. local join  open-addressing build of the received R keys into a table of 2*|R_local| slots, priority
                 = position in the received buffer, home slot = (key >> log2 G) & mask under the low-bit split
                 (the shard bits are the same for every local key), key & mask under the range split; probe
                 with the received S keys
                                                                     -> hj_build_keys_dev / hj_probe_keys_dev
  6. counters    one all_reduce(sum) of {conflicts, matches, sums}

Result semantics: shard g's table holds the tuples whose destination digit is g, inserted in GLOBAL
input order with the reference's probe budget; the test suite restates exactly that on the CPU and
compares bit-exactly (tests/test_sharded_gloo.py). For G = 1 this is the single-GPU operator. For unique keys
(sorted / shuffle / local_shuffle) the totals equal the single-table result (conflicts 0,
matches |R|); for duplicate keys they are the radix-partitioned variant of it, a different but
equally deterministic number (linear-probe neighbourhoods differ once the table is split). Sizes:
2^31 tuples per rank and relation, at most 2^32 received per rank.

The compute engine is injected: HipShardEngine (below) is the product path and the only engine in
this package; the CPU tests inject a checker-backed engine of their own to exercise this file's
exchange logic under gloo without a GPU.
"""
import time

import numpy as np


def _log2(n):
    l = n.bit_length() - 1
    if n <= 0 or (1 << l) != n:
        raise ValueError(f"number of shards must be a power of two, got {n}")
    return l


class HipShardEngine:
    """Product engine: HIP kernels through the C ABI, torch only owns the device buffers."""

    def __init__(self, hj, torch, device_index, build_variant=0):
        self.hj, self.torch = hj, torch
        self.dev = torch.device("cuda", device_index)
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        self.ctx = hj.HashJoinContext(device_index, stream=stream)
        self.build_variant = build_variant
        self._reserved = None

    def close(self):
        self.ctx.close()

    def empty_keys(self, n):
        """buffer of n 32-bit keys (int32 holds the bit pattern; +4 elements so 16-byte sweeps stay inside)"""
        return self.torch.empty(int(n) + 4, dtype=self.torch.int32, device=self.dev)[: int(n)]

    def histogram(self, t, n_shards, mode=0):
        counts = self.torch.zeros(n_shards, dtype=self.torch.int64, device=self.dev)
        self.ctx.shard_histogram(t.data_ptr(), t.numel(), n_shards, counts.data_ptr(), mode)
        return counts

    def scatter(self, t, n_shards, counts, mode=0):
        out = self.empty_keys(t.numel())
        self.ctx.shard_scatter(t.data_ptr(), t.numel(), n_shards, counts.data_ptr(), out.data_ptr(), mode)
        return out

    def max_key(self, t):
        return int(t.max().item()) if t.numel() else 0

    def reserve(self, table_size, max_r, max_s):
        key = (table_size, max_r, max_s)
        if self._reserved != key:
            # the queue of the LDS-window build is sized from rSize: reserve for the larger of the
            # nominal share and what actually arrived
            r = table_size // 2
            while r + r // 8 < max_r:       # hj_reserve keeps 1/8 headroom for uneven shards
                r *= 2
            self.ctx.reserve("atomic", r, max_s, buildVariant=self.build_variant)
            self._reserved = key

    def build(self, r_keys, home_shift, table_size):
        self.ctx.build_keys(r_keys.data_ptr(), r_keys.numel(), home_shift, table_size)

    def probe(self, s_keys):
        self.ctx.probe_keys(s_keys.data_ptr(), s_keys.numel())

    def build_tuples(self, r_tuples):
        """the single-GPU operator on a rank's own 8-byte tuples (table of 2 * len slots, no home shift)"""
        self.ctx.build(r_tuples.data_ptr(), r_tuples.numel())

    def probe_tuples(self, s_tuples):
        self.ctx.probe(s_tuples.data_ptr(), s_tuples.numel())

    def set_check(self, n_shards, mode=0, shard_id=0):
        self.ctx.set_shard_check(n_shards, mode, shard_id)

    def foreign(self):
        """tuples of the last build + probes whose destination was another shard (waits for the stream)"""
        return self.ctx.fetch()["foreignTuples"]

    def finish(self):
        self.ctx.checksums()
        r = self.ctx.fetch()
        return {k: r[k] for k in ("conflicts", "totalMatches", "inputSum", "tableSumFull", "conflictSum",
                                  "buildVariant", "buildDeferred", "build_us", "probe_us", "clear_us", "buildPhaseA_us")}

    def sync(self):
        self.torch.cuda.synchronize(self.dev)


class ShardedJoin:
    """Runs the sharded join for one rank. `dist` is torch.distributed (or None when world == 1)."""

    ONE_BASED = 0x100                   # HJ_SHARD_ONE_BASED: split on (key - 1)

    def __init__(self, engine, torch, dist, rank, world, split="low", max_key=None, exchange="p2p"):
        """split: which key bits pick the destination (both relations use the same).
             "low"   key & (G-1): balanced whatever the keys are, but moves (G-1)/G of a relation that is held as
                     contiguous key ranges
             "high"  the top log2 G bits of the key domain [1, max_key] = a range split: such a relation mostly
                     stays where it is
             "auto"  "high" if at least 3/4 of the tuples of every rank would stay under it, else "low" -- the
                     reference's idea of exploiting locality where the data has it, applied to the exchange
           max_key: upper bound of the keys (DataGen: the relation size); found by one all-reduce if not given.
           exchange: "p2p" = one batch of direct pairwise isend/irecv per relation in messages of <= max_msg_tuples;
                     "a2a" = ONE all_to_all_single per relation (the collective BASELINE config 4 names) whenever no
                     per-peer message of the step exceeds max_msg_tuples on any rank, else the p2p batch for that step
                     (this RCCL build delivered half of a >= 2 GiB per-peer message, tools/dbg/a2a_check.py). Both
                     deliver the same bytes to the same places; which is faster on xGMI is for the first multi-GPU
                     run to say (bench.py --exchange)."""
        self.e, self.torch, self.dist, self.rank, self.world = engine, torch, dist, rank, world
        if split not in ("low", "high", "auto"):
            raise ValueError(f"split must be low, high or auto, got {split!r}")
        if exchange not in ("p2p", "a2a"):
            raise ValueError(f"exchange must be p2p or a2a, got {exchange!r}")
        self.split, self.max_key, self.exchange = split, max_key, exchange
        self.mode = None                # decided on the first step: digit position (| ONE_BASED)
        self.shift = 0                  # home-slot shift of the local tables: log2 G for the low-bit split
        self.last = {}
        self._keep = None
        self._in_place_hint = False     # the previous step found every tuple already on its rank
        self._step_max_msg = 0          # largest single message of the current step over all ranks (keys)
        self.last_exchange_form = None  # "all_to_all_single" / "batch_isend_irecv" of the last exchange that ran

    def _all_reduce_scalar(self, v, op):
        if self.dist is None or self.world == 1:
            return v
        t = self.torch.tensor([v], dtype=self.torch.int64)
        t = t.to(self.e.dev) if hasattr(self.e, "dev") else t
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
        return int(t.item())

    def _decide_split(self, r_local, s_local):
        gbits = _log2(self.world)
        low = (0, gbits)
        if self.world == 1 or self.split == "low":
            self.mode, self.shift = low
            return
        mk = self.max_key
        if mk is None:
            mk = self._all_reduce_scalar(max(self.e.max_key(r_local), self.e.max_key(s_local)), "MAX")
        digit = max((max(mk, 1) - 1).bit_length() - gbits, 0)          # top log2 G bits of (key - 1)
        high = (digit | self.ONE_BASED, 0)
        if digit == 0:
            self.mode, self.shift = low
        elif self.split == "high":
            self.mode, self.shift = high
        else:
            stay = int(self.e.histogram(r_local, self.world, high[0])[self.rank].item()) + \
                   int(self.e.histogram(s_local, self.world, high[0])[self.rank].item())
            total = r_local.numel() + s_local.numel()
            ok = 1 if 4 * stay >= 3 * total else 0
            self.mode, self.shift = high if self._all_reduce_scalar(ok, "MIN") else low

    def _exchange_counts(self, cnt_r, cnt_s):
        """One small all_to_all_single: row d of `both` = (R tuples, S tuples I send to rank d, my largest message).
        Returns my send counts, my receive counts, and the largest single message of the step over ALL ranks (every rank
        computes the same value from what it received, so the choice between the two exchange forms needs no
        collective of its own)."""
        big = self.torch.maximum(cnt_r.max(), cnt_s.max()).reshape(1).expand(self.world)
        both = self.torch.stack([cnt_r, cnt_s, big], dim=1).contiguous()                   # [dest][R, S, big]
        recv = self.torch.empty_like(both)
        if self.dist is not None:
            self.dist.all_to_all_single(recv.view(-1), both.view(-1))
        else:
            recv.copy_(both)
        send = both.cpu().tolist()
        got = recv.cpu().tolist()
        self._step_max_msg = max(g[2] for g in got)
        return [s[0] for s in send], [s[1] for s in send], [g[0] for g in got], [g[1] for g in got]

    # Largest single message, in keys. RCCL (ROCm 7.0 wheel of torch 2.10) delivered only half of an
    # all_to_all_single whose per-peer message reached 2 GiB (tools/dbg/a2a_check.py: 2^27 int64 fine, 2^28
    # half missing), so the exchange is issued as point-to-point messages of at most 512 MiB.
    max_msg_tuples = 1 << 27

    def _exchange_async(self, send, send_counts, recv_counts):
        """Starts the exchange of one relation and returns (output tensor, work handles). With NCCL/RCCL the transfers
        run on the backend's stream behind the kernels already enqueued on the current stream, and work.wait() only
        makes the current stream wait (no host block). The receiver lays the pieces out in source-rank order."""
        out = self.e.empty_keys(sum(recv_counts))
        if self.dist is None or self.world == 1:
            out.copy_(send)
            return out, []
        if self.exchange == "a2a" and self._step_max_msg <= self.max_msg_tuples:
            # the scatter output is grouped by destination, the receive buffer by source: exactly all_to_all_single's
            # layout, one collective call, my own share copied inside it
            w = self.dist.all_to_all_single(out, send, output_split_sizes=list(recv_counts),
                                            input_split_sizes=list(send_counts), async_op=True)
            self.last_exchange_form = "all_to_all_single"
            return out, [w]
        # direct pairwise sends (every peer at once, so all xGMI links carry traffic; no ring), batched into one group
        self.last_exchange_form = "batch_isend_irecv"
        so = [0]
        for c in send_counts:
            so.append(so[-1] + c)
        ro = [0]
        for c in recv_counts:
            ro.append(ro[-1] + c)
        me = self.rank
        if recv_counts[me]:
            out[ro[me]:ro[me + 1]].copy_(send[so[me]:so[me + 1]])           # my own share never leaves the GPU
        works = []
        step = self.max_msg_tuples
        ops = []
        for off in range(1, self.world):                                  # staggered peer order
            d = (me + off) % self.world
            for k in range(0, send_counts[d], step):
                ops.append(self.dist.P2POp(self.dist.isend, send[so[d] + k: so[d] + min(k + step, send_counts[d])], d))
            src = (me - off) % self.world
            for k in range(0, recv_counts[src], step):
                ops.append(self.dist.P2POp(self.dist.irecv, out[ro[src] + k: ro[src] + min(k + step, recv_counts[src])], src))
        if ops:
            works = self.dist.batch_isend_irecv(ops)
        return out, works

    def _try_in_place(self, r_local, s_local, table_size):
        """Optimistic step after one that moved nothing: join the pieces in place with the shard check riding on the
        build and probe kernels (no histogram pass, no split); accept if no rank saw a foreign tuple, else report
        False and let the caller run the full path (the table is rebuilt there)."""
        # _in_place_hint, world and shift are the same on every rank (they come out of collectives); whether MY piece
        # fits the table is not -- a rank whose piece does not still takes part in the all-reduce below (it reports a
        # foreign tuple), so every rank makes the same collective calls whatever its piece looks like
        if not (self._in_place_hint and self.world > 1 and self.shift == 0):
            return False
        e = self.e
        fits = table_size == 2 * r_local.numel()
        foreign = 1
        if fits:
            e.set_check(self.world, self.mode, self.rank)
            e.reserve(table_size, r_local.numel(), s_local.numel())
            e.build_tuples(r_local)
            e.probe_tuples(s_local)
            foreign = e.foreign()
            e.set_check(0)
        if self._all_reduce_scalar(foreign, "SUM") != 0:
            self._in_place_hint = False
            return False
        self.last = {"sent_r": 0, "sent_s": 0, "recv_r": r_local.numel(), "recv_s": s_local.numel(),
                     "split": f"high key bits (range split, digit at bit {self.mode & 0xFF}); pieces joined in place, "
                              "membership checked inside build and probe"}
        return True

    def step(self, r_local, s_local, table_size):
        """One build+probe over this rank's pieces (rank g holds the g-th contiguous piece of each relation).
        Everything is enqueued; call result() to sync. The exchange of R overlaps the split of S, and the
        exchange of S overlaps the local build."""
        e = self.e
        # the previous step's buffers go back to the allocator first (its kernels precede ours in stream order, its
        # transfers are done -- we waited on them): every step after the first reuses the same blocks, no hipMalloc
        self._keep = None
        if self.mode is None:
            self._decide_split(r_local, s_local)
        if self._try_in_place(r_local, s_local, table_size):
            return
        cnt_r = e.histogram(r_local, self.world, self.mode)
        cnt_s = e.histogram(s_local, self.world, self.mode)
        send_r, send_s, recv_r, recv_s = self._exchange_counts(cnt_r, cnt_s)
        split_name = "low key bits" if self.mode == 0 else f"high key bits (range split, digit at bit {self.mode & 0xFF})"
        moved = sum(send_r) - send_r[self.rank] + sum(send_s) - send_s[self.rank]
        # world and shift are global, the piece's size is not: it goes INTO the reduced value instead of deciding
        # whether this rank calls the collective (round-1 ADVICE: a mismatched collective with ragged pieces)
        stay = self.world > 1 and self.shift == 0 and \
            self._all_reduce_scalar(moved + (0 if table_size == 2 * r_local.numel() else 1), "SUM") == 0
        if stay:
            # a range split under which every tuple already sits on its rank: nothing to regroup, nothing to send --
            # the rank's piece IS its shard, in input order
            e.reserve(table_size, r_local.numel(), s_local.numel())
            e.build_tuples(r_local)
            e.probe_tuples(s_local)
            self.last = {"sent_r": 0, "sent_s": 0, "recv_r": r_local.numel(), "recv_s": s_local.numel(),
                         "split": split_name + "; no tuple had to move: pieces joined in place"}
            self._in_place_hint = True          # next step: try it straight away, checked inside build and probe
            return
        out_r = e.scatter(r_local, self.world, cnt_r, self.mode)                # keys, grouped by destination
        got_r, work_r = self._exchange_async(out_r, send_r, recv_r)
        out_s = e.scatter(s_local, self.world, cnt_s, self.mode)
        got_s, work_s = self._exchange_async(out_s, send_s, recv_s)
        e.reserve(table_size, got_r.numel(), got_s.numel())
        for w in work_r:
            w.wait()
        e.build(got_r, self.shift, table_size)
        for w in work_s:
            w.wait()
        e.probe(got_s)
        self.last = {"sent_r": sum(send_r) - send_r[self.rank], "sent_s": sum(send_s) - send_s[self.rank],
                     "recv_r": got_r.numel(), "recv_s": got_s.numel(), "split": split_name}
        self._keep = (out_r, out_s, got_r, got_s)   # alive until the stream has consumed them

    def step_phases(self, r_local, s_local, table_size):
        """One step with the phases run one after the other (device synchronised and all ranks at a barrier between
        them) and timed on the host: what the overlapped step() hides is visible here. Returns milliseconds per phase;
        exchange_* is the time from all ranks having their send buffers ready to all of this rank's pieces having
        arrived. Only for a split that really exchanges (not the in-place shortcut)."""
        e, ph = self.e, {}
        self._keep = None
        if self.mode is None:
            self._decide_split(r_local, s_local)

        def lap(name, t0):
            e.sync()
            ph[name] = ph.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
            if self.dist is not None and self.world > 1:
                self.dist.barrier()
            return time.perf_counter()

        e.sync()
        t = lap("_", time.perf_counter())
        cnt_r = e.histogram(r_local, self.world, self.mode)
        cnt_s = e.histogram(s_local, self.world, self.mode)
        t = lap("histogram", t)
        send_r, send_s, recv_r, recv_s = self._exchange_counts(cnt_r, cnt_s)
        t = lap("counts", t)
        out_r = e.scatter(r_local, self.world, cnt_r, self.mode)
        out_s = e.scatter(s_local, self.world, cnt_s, self.mode)
        t = lap("split", t)
        got_r, work = self._exchange_async(out_r, send_r, recv_r)
        for w in work:
            w.wait()
        t = lap("exchange_r", t)
        got_s, work = self._exchange_async(out_s, send_s, recv_s)
        for w in work:
            w.wait()
        t = lap("exchange_s", t)
        e.reserve(table_size, got_r.numel(), got_s.numel())
        e.build(got_r, self.shift, table_size)
        t = lap("build", t)
        e.probe(got_s)
        t = lap("probe", t)
        del ph["_"]
        self.last = {"sent_r": sum(send_r) - send_r[self.rank], "sent_s": sum(send_s) - send_s[self.rank],
                     "recv_r": got_r.numel(), "recv_s": got_s.numel(),
                     "split": "low key bits" if self.mode == 0 else f"high key bits (range split, digit at bit {self.mode & 0xFF})"}
        self._keep = (out_r, out_s, got_r, got_s)
        return ph

    def result(self):
        r = self.e.finish()
        keys = ("conflicts", "totalMatches", "inputSum", "tableSumFull", "conflictSum", "buildDeferred")
        t = self.torch.tensor([r[k] for k in keys], dtype=self.torch.int64)
        if self.dist is not None:
            t = t.to(self.e.dev) if hasattr(self.e, "dev") else t
            self.dist.all_reduce(t)
            t = t.cpu()
        out = dict(zip(keys, (int(x) for x in t.tolist())))
        out["local"] = r
        out["exchange"] = dict(self.last)
        return out


def _local_roofline(res):
    """Roofline of the dominant LOCAL kernel on rank 0 in the last step (the exchange is xGMI-bound and reported
    beside it as ms_per_step - local kernels): k_build_own over the received keys, 4 B read + 8 B slot written per
    R tuple; only when the LDS-window build ran."""
    loc, ex = res["local"], res["exchange"]
    if loc["buildVariant"] not in (2, 3) or not loc["buildPhaseA_us"]:
        return None
    nbytes = 12.0 * ex["recv_r"]
    gbps = nbytes / (loc["buildPhaseA_us"] * 1e-6) / 1e9
    kname = "k_build_wave" if loc["buildVariant"] == 3 else "k_build_own"
    return {"bound": "hbm", "kernel": kname + "<keys> (rank 0, last step)", "achieved": gbps, "peak": 8000.0,
            "unit": "GB/s", "frac": gbps / 8000.0, "traffic": None, "algorithmic_bytes_per_launch": nbytes,
            "launch_us": loc["buildPhaseA_us"]}


def rank_key_range(world, n, rank):
    """(lo, width): rank `rank` of `world` draws its n keys from (lo, lo + width]. While the whole relation fits the
    32-bit key space that is (rank*n, (rank+1)*n]; otherwise the rank-th of `world` equal cuts of the key space (the
    last one ends at 2^32 - 1), so that the range split's digit ((key - 1) >> d) names the owner of every key."""
    wrap = (1 << 32) - 1
    if world * n <= wrap:
        return rank * n, n
    cut = (wrap + world) // world                                 # 2^32 / world for a power-of-two world
    return rank * cut, min(cut, wrap - rank * cut)


def squeeze_into_range(v, n, lo, width, np):
    """values in [1, n] -> keys in (lo, lo + width], order kept (width < n: neighbouring values share a key)"""
    v = v - np.uint64(1)
    if width != n:
        v = (v * np.uint64(width)) // np.uint64(n)
    return v + np.uint64(lo + 1)


def exchange_selfcheck(job, n_step_keys):
    """Pre-flight of the exchange, before anything is timed: both forms (`all_to_all_single`, the pairwise batch) move a
    known pattern at 1 MiB per peer and at the step's own per-peer size through ShardedJoin._exchange_async -- the very
    function the steps use -- and EVERY received element is compared with what its sender must have put there. The first
    multi-rank contact of this code with RCCL happens here, where a wrong byte costs a line in the report, not a wrong
    benchmark (all_to_all_single of this RCCL build dropped half of a >= 2 GiB per-peer message at world 1,
    tools/dbg/a2a_check.py). Returns {size: {form: {"ok", "ms"}}}, identical on every rank (flags MIN-reduced, times
    MAX-reduced). A form is skipped at a size where a per-peer message would exceed ShardedJoin.max_msg_tuples for
    all_to_all_single (the step falls back to the batch there anyway)."""
    torch, dist, world, rank, e = job.torch, job.dist, job.world, job.rank, job.e
    report = {}
    saved = job.exchange, job._step_max_msg, job.last_exchange_form
    sizes = []
    for name, per_peer in (("1MiB_per_peer", 1 << 18), ("step_size", int(n_step_keys))):
        if per_peer > 0 and all(per_peer != s for _, s in sizes):
            sizes.append((name, per_peer))
    for name, per_peer in sizes:
        report[name] = {"keys_per_peer": per_peer}
        for form in ("a2a", "p2p"):
            if form == "a2a" and per_peer > job.max_msg_tuples:
                report[name][form] = {"ok": None, "ms": None, "skipped": "per-peer message above max_msg_tuples: the step uses the batch"}
                continue
            send = e.empty_keys(per_peer * world)
            # element i of the piece rank s sends to rank d: (s * world + d) * 40503 + i, wrapped to 31 bits
            i = torch.arange(per_peer, dtype=torch.int64, device=send.device)
            for d in range(world):
                send[d * per_peer:(d + 1) * per_peer] = (((rank * world + d) * 40503 + i) & 0x7FFFFFFF).to(torch.int32)
            counts = [per_peer] * world
            job.exchange, job._step_max_msg = form, per_peer
            ok, best = 1, None
            for _ in range(2):                                   # the second pass is the timed one (first: connections come up)
                e.sync()
                dist.barrier()
                t0 = time.perf_counter()
                got, works = job._exchange_async(send, counts, counts)
                for w in works:
                    w.wait()
                e.sync()
                best = (time.perf_counter() - t0) * 1e3
                for s_ in range(world):
                    want = (((s_ * world + rank) * 40503 + i) & 0x7FFFFFFF).to(torch.int32)
                    if not torch.equal(got[s_ * per_peer:(s_ + 1) * per_peer], want):
                        ok = 0
                del got
            used = job.last_exchange_form
            ok = job._all_reduce_scalar(ok, "MIN")
            ms = job._all_reduce_scalar(int(best * 1e6), "MAX") / 1e6
            report[name][form] = {"ok": bool(ok), "ms": ms, "form": used,
                                  "GBps_out_per_rank": 4.0 * per_peer * (world - 1) / (ms * 1e-3) / 1e9 if ms else None}
            del send
    job.exchange, job._step_max_msg, job.last_exchange_form = saved
    return report


def pick_exchange(report):
    """"a2a" if all_to_all_single delivered every element at every size it was tried and was not slower than the batch
    at the step's size, else "p2p"; a report in which the batch itself failed raises (nothing to fall back to)."""
    for size in report.values():
        if size["p2p"]["ok"] is False:
            raise RuntimeError(f"exchange self-check: the pairwise batch delivered wrong data: {report}")
    a2a_ok = all(size["a2a"]["ok"] is True for size in report.values())
    last = list(report.values())[-1]
    if a2a_ok and last["a2a"]["ms"] <= last["p2p"]["ms"]:
        return "a2a"
    return "p2p"


def bench_sharded(args, torch, dist, hj, rank, world, local_rank, engine=None, device=None):
    """bench.py's N > 1 leg = BASELINE config 4: radix join across the GPUs of one node with an all-to-all partition
    exchange. Default = WEAK scaling: every rank holds 2^log2n tuples of R and of S (the N=1 workload per GPU; 8 GPUs x
    2^30 = config 4); --strong keeps the N=1 total and splits it.

    `value` is measured with the LOW-BIT split (destination = key & (G-1), HASH_BIT_MODULO of
    mc/src/parallel_radix_join.c:59): (G-1)/G of every relation really crosses xGMI in every step. The range split
    (--split high / auto), under which the bench's contiguous near-sorted pieces hardly move and the ranks end up
    joining in place, is timed for a few steps beside it and reported as `other_split` -- it says what exploiting the
    data's locality in the exchange is worth, it is not the config-4 number. `phase_ms` / `exchange_ms` / `local_ms`
    come from extra steps whose phases run one after the other (step_phases): exchange and local time separately, as
    SURVEY.md 8e asks.

    Globally the relation is the near-sorted one the reference generates, cut into contiguous pieces: rank g's
    piece = DataGen over its own key range, drawn piecewise (a rank cannot afford the serial rand() stream of its
    neighbours). Keys are 32 bits: while world * n <= 2^32 - 1 rank g owns the keys (g*n, (g+1)*n]; beyond that
    (8 x 2^30 tuples, 2^32 keys) the key space is cut into `world` equal contiguous ranges and a rank's n draws are
    squeezed into its range -- every key about world*n / 2^32 times, duplicates adjacent, as in a globally sorted
    relation with more tuples than keys.

    engine / device: the CPU tests pass a checker-backed engine and "cpu" to run this very function under gloo."""
    strip = _log2(world)
    n = (1 << args.log2n) >> (strip if args.strong else 0)          # tuples per rank and relation
    window = args.shuffle_range
    wrap = (1 << 32) - 1
    dev = device or f"cuda:{local_rank}"
    lo, width = rank_key_range(world, n, rank)
    R = hj.generate_data(args.dist, n, n, window)                    # values in [1, n] (uniform and the unique-key kinds)

    def to_range(v):
        if args.dist == "random":                                    # 31-bit random keys: no range to speak of
            return v
        return squeeze_into_range(v, n, lo, width, np)
    r_local = torch.from_numpy(to_range(R).view("int64")).to(dev)
    # S as main.cpp:91-97 builds it: sorted 1..N on the same key range (for `random`: R itself)
    S = R.copy() if args.dist == "random" else to_range(np.arange(1, n + 1, dtype=np.uint64))
    s_local = torch.from_numpy(S.view("int64")).to(dev)
    del R, S
    eng = engine if engine is not None else HipShardEngine(hj, torch, local_rank, build_variant=args.build_variant)
    table_size = 2 * n
    total_keys = min(world * n, wrap)                                # the keys lie in [1, relation size] (or the whole key space)
    exchange = getattr(args, "exchange", "auto")
    # pre-flight: the first thing the ranks do together is an exchange whose every element is checked
    selfcheck = None
    if world > 1:
        probe_job = ShardedJoin(eng, torch, dist, rank, world, split="low", max_key=total_keys, exchange="p2p")
        selfcheck = exchange_selfcheck(probe_job, (n + world - 1) // world)
        if exchange == "auto":
            exchange = pick_exchange(selfcheck)
        elif exchange == "a2a" and not all(sz["a2a"]["ok"] is not False for sz in selfcheck.values()):
            raise RuntimeError(f"--exchange a2a: all_to_all_single failed the self-check: {selfcheck}")
        else:
            pick_exchange(selfcheck)                                 # raises if even the batch is wrong
    elif exchange == "auto":
        exchange = "p2p"

    def new_job(split):
        return ShardedJoin(eng, torch, dist, rank, world, split=split,
                           max_key=None if args.dist == "random" else total_keys, exchange=exchange)

    def timed(split, steps, warmup):
        job = new_job(split)
        for _ in range(warmup):
            job.step(r_local, s_local, table_size)
        eng.sync(); dist.barrier(); eng.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            job.step(r_local, s_local, table_size)
        eng.sync(); dist.barrier(); eng.sync()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return job, float(t.item())

    job, dt = timed(args.split, args.steps, args.warmup)
    res = job.result()
    total = 2 * n * world
    # exchange and local time separately: a few steps with the phases run one after the other, maximum over ranks
    phase_ms = None
    if world > 1 and "in place" not in res["exchange"]["split"]:
        pj = new_job(args.split)
        runs = [pj.step_phases(r_local, s_local, table_size) for _ in range(3)][1:]      # the first one warms up
        names = list(runs[0])
        t = torch.tensor([[r_[k] for k in names] for r_ in runs], dtype=torch.float64, device=dev).min(dim=0).values
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        phase_ms = dict(zip(names, (float(x) for x in t.tolist())))
    # the same join with the other split, a few steps, for the record (never part of `value`)
    alt = None
    if world > 1 and not getattr(args, "no_other_split", False):
        other = "auto" if job.mode == 0 else "low"
        k = max(1, min(3, args.steps))
        ajob, adt = timed(other, k, 1)
        ares = ajob.result()
        alt = {"split": ares["exchange"]["split"], "steps": k, "ms_per_step": adt / k * 1e3,
               "mtuples_per_s": total * k / adt / 1e6, "sent_r": ares["exchange"]["sent_r"], "sent_s": ares["exchange"]["sent_s"],
               "conflicts": ares["conflicts"], "totalMatches": ares["totalMatches"]}
    unique_domain = world * n <= wrap
    unique = unique_domain and args.dist in ("sorted", "shuffle", "local_shuffle")
    line = {
        "metric": "Mtuples/sec build+probe, |R|=|S|=1B uint32, uniform vs local_shuffle",
        "value": total * args.steps / dt / 1e6, "unit": "Mtuples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if args.strong else "weak",
        "vs_baseline": None, "dtype": "u64 tuples (u32 key), integer",
        "data": "synthetic (DataGen restatement per rank on its own key range)",
        "config": {"workload": f"radix-sharded open-addressing build+probe over {world} GPUs, |R|=|S|={n} per GPU "
                               f"({n * world} in total), dataDistr={args.dist} W={window}, every rank holding one contiguous "
                               f"piece of the near-sorted relations; destination = {res['exchange']['split']} (--split "
                               f"{args.split}); step = destination histogram + stable split to 32-bit keys + all-to-all "
                               f"(R and S, overlapped with split and build; --exchange {exchange}) + local table build/probe",
                   "algo": "atomic", "rSize": n * world,
                   "sSize": n * world, "per_gpu_rSize": n, "dataDistr": args.dist, "shuffleRange": window,
                   "parallelism": f"radix{world}"},
        "result": {k: res[k] for k in ("conflicts", "totalMatches", "inputSum", "buildDeferred")},
        "checks": {"matches_plus_conflicts_eq_rSize": (res["totalMatches"] + res["conflicts"] == n * world) if unique_domain else None,
                   "unique_keys_all_match": (res["totalMatches"] == n * world) if unique else None,
                   "tableSum_plus_conflictSum_eq_inputSum": res["tableSumFull"] + res["conflictSum"] == res["inputSum"]},
        "a2a_selfcheck": selfcheck, "exchange_chosen": exchange,
        "exchange": {**res["exchange"], "form": job.last_exchange_form,
                     "bytes_sent_per_rank_per_step": 4 * (res["exchange"]["sent_r"] + res["exchange"]["sent_s"])},
        "phase_ms": phase_ms,
        "exchange_ms": (phase_ms["exchange_r"] + phase_ms["exchange_s"]) if phase_ms else None,
        "local_ms": (sum(v for k_, v in phase_ms.items() if not k_.startswith("exchange"))) if phase_ms else None,
        "local_kernel_us": {k: res["local"][k] for k in ("clear_us", "build_us", "probe_us")},
        "local_build_variant": res["local"]["buildVariant"],
        "other_split": alt,
        "roofline": _local_roofline(res), "cpu_baseline": None,
    }
    if engine is None:
        eng.close()
    return line
