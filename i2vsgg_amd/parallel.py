"""One process per GPU over torch.distributed (backend "nccl" = RCCL on ROCm, xGMI inside a node).

The path shards by independent frames: each rank runs the whole forward/backward on
its own frames with the loss scaled by 1/world, and the only exchange is ONE sum all-reduce of the
trainable-parameter gradients per step.  Large gradients (the 822 MB ``vrd.fc6`` weight) are reduced
in place, tensor by tensor, as asynchronous collectives; tensors below ``SMALL_BYTES`` are packed into
one flat bucket so that their latency is paid once.  xGMI is point-to-point (7 links x ~153 GB/s per
GPU), so a few large collectives beat many small ones.
"""
import os

import torch
import torch.distributed as dist

SMALL_BYTES = 1 << 20


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def exchange_enabled():
    """True when gradients are exchanged: world_size > 1, or a 1-rank group forced with I2V_FORCE_EXCHANGE=1
    (rehearses the RCCL path and the pipelined step schedule on a single GPU)."""
    if world_size() > 1:
        return True
    return os.environ.get("I2V_FORCE_EXCHANGE") == "1" and dist.is_available() and dist.is_initialized()


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def any_rank(flag, device="cpu"):
    """True on every rank when ``flag`` is true on at least one (a collective: every rank must call it).  Single process: ``flag``."""
    if world_size() <= 1:
        return bool(flag)
    t = torch.tensor([1.0 if flag else 0.0], device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return bool(t.item() > 0)


def capture_kwargs():
    """Extra arguments for ``torch.cuda.graph`` while a process group exists.  ProcessGroupNCCL's watchdog thread polls the
    events of collectives still on its list (those of the eager warm-up steps) with hipEventQuery; under the default GLOBAL
    capture mode a query from ANY thread while a capture is running is an error, the watchdog thread throws, nothing catches it
    and the process aborts (round 5: the 1-rank RCCL rehearsal of the instance_styleD step died so, SIGABRT inside its capture;
    the relation step had been getting away with it on timing).  THREAD_LOCAL confines the check to the capturing thread.
    ``wait_for_collectives`` below empties the watchdog's list first, which is the other half."""
    if dist.is_available() and dist.is_initialized():
        return {"capture_error_mode": "thread_local"}
    return {}


WATCHDOG_PERIOD_S = 0.1       # ProcessGroupNCCL::kWatchdogThreadSleepMillis (torch/csrc/distributed/c10d/ProcessGroupNCCL.hpp): one pass per 100 ms


def wait_for_collectives(device):
    """Before a capture: every collective launched so far has finished AND has left the watchdog's work list.

    The watchdog thread polls the end events of the works on its list (``hipEventQuery``) and drops a finished work on its
    next pass.  While a capture runs that query can abort the process -- the watchdog throws, nothing catches it: under the
    GLOBAL capture mode always (round 5), and under THREAD_LOCAL (``capture_kwargs``) still sometimes: the works of the warm-up
    steps were recorded on RCCL's own stream, which the captured collectives pull INTO the capture.  Round 6 tried the
    synchronisation alone -- THREAD_LOCAL should make the query legal -- and the full GPU suite aborted in the instance_styleD
    rehearsal after three clean runs (profiles/r06_watchdog_abort.txt).  So the list must be EMPTY when the capture starts.
    torch offers no call for that (no ``_wait_for_pending_works`` on this build; ``Work.wait()`` is a stream wait, and
    ``Work.is_completed()`` is true long before the watchdog's copy is gone), which leaves the watchdog's own clock: after a
    device synchronisation every work is complete, and one full pass later -- two periods, to be safe against a pass that
    was under way -- the list is empty.  Works issued DURING a capture are never put on the list (ProcessGroupNCCL skips
    ``workEnqueue`` while capturing)."""
    if dist.is_available() and dist.is_initialized() and device.type == "cuda":
        import time
        torch.cuda.synchronize(device)
        time.sleep(3 * WATCHDOG_PERIOD_S)


def init_from_env(backend=None):
    """Initialise from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rk = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rk)))
    use_cuda = torch.cuda.is_available()
    device = torch.device("cuda", local % max(torch.cuda.device_count(), 1)) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    if (world > 1 or os.environ.get("I2V_FORCE_EXCHANGE") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if use_cuda and (backend or "nccl") == "nccl":
            kw["device_id"] = device
        dist.init_process_group(backend or ("nccl" if use_cuda else "gloo"), rank=rk, world_size=world, **kw)
    return rk, world, device


def shard_frames(n_global, rk=None, world=None):
    """Contiguous split of the global frame batch: returns (start, stop) of this rank."""
    rk = rank() if rk is None else rk
    world = world_size() if world is None else world
    per, rem = divmod(n_global, world)
    start = rk * per + min(rk, rem)
    return start, start + per + (1 if rk < rem else 0)


def all_reduce_grads_start(params, group=None):
    """Launch the sum-all-reduce of every ``p.grad`` (loss was pre-scaled by 1/world) and return a token for
    ``all_reduce_grads_finish``.  The collectives run on RCCL's own stream behind everything already queued
    on the current stream; work queued on the current stream AFTER this call overlaps them."""
    if not exchange_enabled():
        return None
    # a FIXED list on every rank: a parameter that received no gradient here (an unused branch on this rank's batch)
    # contributes zeros -- skipping it would give the ranks different collective sequences (a hang or mixed-up sums)
    grads = []
    for p in params:
        if is_local(p):
            continue
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        grads.append(p.grad)
    big = [g for g in grads if g.numel() * g.element_size() >= SMALL_BYTES]
    small = [g for g in grads if g.numel() * g.element_size() < SMALL_BYTES]
    # largest first: the 822 MB fc6 gradient dominates the exchange
    handles = [dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group, async_op=True)
               for g in sorted(big, key=lambda t: -t.numel())]
    flat = None
    if small:
        flat = torch.cat([g.reshape(-1) for g in small])
        handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True))
    return handles, small, flat


def all_reduce_grads_finish(token):
    """Make the current stream wait for the exchange and scatter the small-tensor bucket back."""
    if token is None:
        return
    handles, small, flat = token
    for h in handles:
        h.wait()
    if flat is not None:
        off = 0
        for g in small:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n


def all_reduce_grads(params, group=None):
    """Sum-all-reduce ``p.grad`` of every parameter (loss was pre-scaled by 1/world)."""
    all_reduce_grads_finish(all_reduce_grads_start(params, group))


# ---- the detector step's exchange in buckets, each issued as soon as its gradients exist (round 6) ---------------------------
# nn.DataParallel reduces inside backward() (trainval_net_instance_styleD_bilinear.py:200-201, :324-333); rounds 1-5 issued ONE
# all-reduce of the 202 MB after the whole backward and the join of the step's two branches -- all of it exposed.  The
# parameters are cut into a few buckets in BACKWARD order (heads + layer4 + RPN first, layer3 in slices, the early layers
# last); a bucket's sum over the step's branches and its all-reduce are queued on an exchange stream behind the points at
# which each branch has produced the bucket's last gradient, so the exchange of bucket k runs beside the backward of the
# layers below it.  Every rank issues the same collectives in the same order (the bucket list is a function of the parameter
# names only; a parameter without a gradient contributes zeros).

class BucketMarks:
    """Where, on a branch's stream, each bucket's gradients are complete: a parameter's hook calls ``hit(bucket)``, which
    records an event on the current stream (the last record of a bucket stands).  On the CPU (gloo rehearsal) there is no
    stream to mark: the bucket is only noted."""

    def __init__(self, cuda):
        self.cuda, self.events, self.order = bool(cuda), {}, []

    def hit(self, bucket):
        if bucket not in self.events:
            self.order.append(bucket)
        ev = None
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record()
        self.events[bucket] = ev


def exchange_in_buckets(params, buckets, grad_sets, marks=(), stream=None, group=None):
    """``buckets``: lists of indices into ``params`` in backward order.  ``grad_sets``: one gradient tuple per branch (``None``
    where a branch has no gradient for a parameter).  Per bucket, in order: wait (on ``stream``, if given: the current stream is
    expected to BE it) for every branch's mark of the bucket, add the branches' gradients into ``p.grad``, start the all-reduce.
    The caller finishes with ``all_reduce_grads_finish`` on every returned token (``finish_buckets``)."""
    tokens = []
    for k, idx in enumerate(buckets):
        if stream is not None:
            for m in marks:
                ev = m.events.get(k)
                if ev is not None:
                    stream.wait_event(ev)
        first, rest_a, rest_b = [], [], []
        for i in idx:
            gs = [g[i] for g in grad_sets if g[i] is not None]
            params[i].grad = gs[0] if gs else None
            for extra in gs[1:]:
                rest_a.append(gs[0]); rest_b.append(extra)
        if rest_a:
            torch._foreach_add_(rest_a, rest_b)
        tokens.append(all_reduce_grads_start([params[i] for i in idx], group))
    return tokens


def finish_buckets(tokens):
    for t in tokens:
        all_reduce_grads_finish(t)


def detector_buckets(names, slices=3):
    """Bucket id per parameter name of the instance_styleD detector, in backward order: 0 = everything above layer3 (the
    layer4 head, classifier, box regressor, netD_pixel, the RPN), 1 .. ``slices`` = layer3 (``RCNN_base.6.<block>``) from its
    last block down, ``slices`` + 1 = the rest (layer2, layer1, netD_style, context heads).  -> (ids, number of buckets)."""
    import re
    blocks = sorted({int(m.group(1)) for n in names for m in [re.match(r"RCNN_base\.6\.(\d+)\.", n)] if m})
    nb = len(blocks)
    ids = []
    for n in names:
        m = re.match(r"RCNN_base\.6\.(\d+)\.", n)
        if m and nb:
            pos = nb - 1 - blocks.index(int(m.group(1)))            # 0 = the last block (first in the backward)
            ids.append(1 + min(slices - 1, pos * slices // nb))
        elif n.startswith(("RCNN_top.", "RCNN_cls_score.", "RCNN_bbox_pred.", "netD_pixel.", "RCNN_rpn.")):
            ids.append(0)
        else:
            ids.append(slices + 1)
    return ids, slices + 2


# ---- tensor parallelism for the one layer where data parallelism is the wrong cut --------------------------------
# vrd.fc6 is a 50176 -> 4096 linear layer (822 MB of fp32 weights) applied to ~128 rows per GPU.  Data-parallel, its
# gradient is 91 % of the bytes of the step's all-reduce, and xGMI is point-to-point: between 2 GPUs that is 822 MB
# over ONE link.  Cut the layer by OUTPUT columns instead: every rank owns 4096/world columns of W (and their momentum),
# sees the rows of ALL ranks (an all-gather of the pooled inputs: 25.7 MB per rank), and the weight gradient of its
# columns is complete locally -- no all-reduce, and the SGD update stays fused into the wgrad epilogue.  The activations
# return to the data-parallel row split through two 2 MB exchanges (forward: my rows of every column shard; backward:
# every rank's gradient for my columns).  Per step and rank: ~30 MB + the 84 MB all-reduce of the other layers, instead
# of 906 MB.  All three collectives are captured inside the head's HIP graph.

def gather_rows(x):
    """(R, C) on every rank -> (world*R, C), rank-major rows.  No autograd (the pooled ROI features carry no gradient)."""
    w = world_size() if dist.is_available() and dist.is_initialized() else 1
    x = x.contiguous()
    if w == 1 and not exchange_enabled():
        return x
    out = x.new_empty((w * x.shape[0],) + tuple(x.shape[1:]))
    dist.all_gather_into_tensor(out, x)
    return out


class ColShardToOwnRows(torch.autograd.Function):
    """h_shard (world*R, C/world) = my columns for everybody's rows  ->  h (R, C) = all columns for MY rows."""

    @staticmethod
    def forward(ctx, h_shard):
        w, rk = world_size(), rank()
        rows, cs = h_shard.shape
        r = rows // w
        ctx.geom = (w, rk, r, cs)
        full = h_shard.new_empty((w, rows, cs))
        dist.all_gather_into_tensor(full.view(w * rows, cs), h_shard.contiguous())
        return full[:, rk * r:(rk + 1) * r, :].permute(1, 0, 2).reshape(r, w * cs)

    @staticmethod
    def backward(ctx, dh):
        w, rk, r, cs = ctx.geom
        allg = dh.new_empty((w, r, w * cs))
        dist.all_gather_into_tensor(allg.view(w * r, w * cs), dh.contiguous())
        return allg[:, :, rk * cs:(rk + 1) * cs].reshape(w * r, cs).contiguous()


def assert_same_rows(n_rows, what="rows"):
    """The column-parallel fc6 exchanges (gather_rows, ColShardToOwnRows) are fixed-size all-gathers: every rank must
    bring the same number of rows.  Host-side check (one small all-gather of an int): call it where shapes are decided
    -- when a step is captured, or per batch in eager mode -- never inside a captured region."""
    if world_size() <= 1:
        return
    t = torch.tensor([int(n_rows)], dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    out = [torch.zeros_like(t) for _ in range(world_size())]
    dist.all_gather(out, t)
    counts = [int(o.item()) for o in out]
    if len(set(counts)) != 1:
        raise RuntimeError("column-parallel vrd.fc6 needs the same number of %s on every rank, got %s: pad the batch to "
                           "a common size or set I2V_TP_FC6=0 (plain data parallelism)" % (what, counts))


def mark_local(p):
    """This parameter's gradient is complete on this rank (a tensor-parallel shard): no exchange, fusable update."""
    p._i2v_local = True
    return p


def is_local(p):
    return getattr(p, "_i2v_local", False)


def max_over_ranks(value, device):
    if world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(device=None):
    if world_size() > 1:
        if device is not None and device.type == "cuda":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()
