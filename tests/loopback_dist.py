"""An in-process stand-in for torch.distributed (TEST ONLY): `world` threads of one process, each driving its own rank of
htm_hashjoin_amd.sharded.ShardedJoin, exchange tensors through a shared hub. It implements exactly the calls ShardedJoin
makes (all_to_all_single, all_reduce, batch_isend_irecv with P2POp/isend/irecv, barrier) so that the PRODUCT engine
(HipShardEngine: HIP kernels through the C ABI) can run the N > 1 code path on the single GPU of the test box -- RCCL
itself refuses two ranks on one device."""
import queue
import threading


class _ReduceOp:
    SUM, MIN, MAX = "SUM", "MIN", "MAX"


class Hub:
    def __init__(self, world):
        self.world = world
        self.bar = threading.Barrier(world)
        self.slots = [None] * world
        self.mail = {(s, d): queue.Queue() for s in range(world) for d in range(world)}


class _Work:
    def wait(self):
        return True


class P2POp:
    def __init__(self, op, tensor, peer):
        self.op, self.tensor, self.peer = op, tensor, peer


class LoopbackDist:
    ReduceOp = _ReduceOp
    P2POp = P2POp
    isend, irecv = "isend", "irecv"

    def __init__(self, hub, rank):
        self.hub, self.rank = hub, rank

    def barrier(self):
        self.hub.bar.wait()

    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None, async_op=False):
        h, w = self.hub, self.hub.world
        isz = list(input_split_sizes) if input_split_sizes is not None else [inp.numel() // w] * w
        osz = list(output_split_sizes) if output_split_sizes is not None else [out.numel() // w] * w
        h.slots[self.rank] = (inp, isz)
        h.bar.wait()
        at = 0
        for s in range(w):
            src, ssz = h.slots[s]
            off = sum(ssz[:self.rank])
            assert ssz[self.rank] == osz[s], (s, self.rank, ssz, osz)
            out[at:at + osz[s]].copy_(src[off:off + osz[s]])
            at += osz[s]
        h.bar.wait()
        return _Work() if async_op else None

    def all_reduce(self, t, op="SUM"):
        h = self.hub
        h.slots[self.rank] = t.clone()
        h.bar.wait()
        vals = [h.slots[s].cpu() for s in range(h.world)]
        acc = vals[0].clone()
        for v in vals[1:]:
            acc = acc + v if op == "SUM" else (acc.minimum(v) if op == "MIN" else acc.maximum(v))
        h.bar.wait()
        t.copy_(acc.to(t.device))

    def batch_isend_irecv(self, ops):
        for o in ops:                      # sends never block ...
            if o.op == "isend":
                self.hub.mail[(self.rank, o.peer)].put(o.tensor)
        for o in ops:                      # ... so the receives of every rank find their messages, in order per pair
            if o.op == "irecv":
                src = self.hub.mail[(o.peer, self.rank)].get(timeout=120)
                assert src.numel() == o.tensor.numel(), (src.numel(), o.tensor.numel())
                o.tensor.copy_(src)
        return [_Work()]
