"""Training helpers with the reference's names (lib/model/utils/net_utils.py)."""
import torch
from torch.utils.data.sampler import Sampler


class sampler(Sampler):
    """Contiguous index blocks of ``batch_size`` in random block order, leftovers last
    (net_utils.py:13-36) -- keeps images of similar aspect ratio (sorted roidb) in one batch.

    ``rank`` / ``world`` (one process per GPU instead of the reference's single-process DataParallel): every rank draws the
    SAME block order (generator seeded with ``seed`` + the epoch count) and keeps the blocks rank, rank + world, ...; the
    leftover images and the blocks that do not fill a round of all ranks are dropped, so every rank sees the same number
    of minibatches per epoch.  The defaults are the reference's sampler."""

    def __init__(self, train_size, batch_size, rank=0, world=1, seed=None):
        self.num_data = train_size
        self.batch_size = batch_size
        self.num_per_batch = train_size // batch_size
        self.leftover = torch.arange(self.num_per_batch * batch_size, train_size).long()
        self.rank, self.world, self.seed, self.epoch = int(rank), int(world), seed, 0

    def __iter__(self):
        gen = None
        if self.seed is not None:
            gen = torch.Generator().manual_seed(int(self.seed) + self.epoch)
        self.epoch += 1
        order = torch.randperm(self.num_per_batch, generator=gen) if gen is not None else torch.randperm(self.num_per_batch)
        if self.world > 1:
            order = order[:self.num_per_batch // self.world * self.world][self.rank::self.world]
        starts = order.view(-1, 1) * self.batch_size
        idx = (starts + torch.arange(self.batch_size).view(1, -1)).view(-1)
        return iter(idx if self.world > 1 else torch.cat((idx, self.leftover), 0))

    def __len__(self):
        if self.world > 1:
            return self.num_per_batch // self.world * self.batch_size
        return self.num_data


class GradReverse(torch.autograd.Function):
    """Gradient reversal layer (net_utils.py:52-61): identity forward, ``-lambd * g`` backward."""

    @staticmethod
    def forward(ctx, x, lambd):
        ctx.lambd = lambd
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * -ctx.lambd, None


def grad_reverse(x, lambd=1.0):
    return GradReverse.apply(x, lambd)


def _smooth_l1_loss(bbox_pred, bbox_targets, bbox_inside_weights, bbox_outside_weights, sigma=1.0, dim=[1]):
    """net_utils.py:122-136.  On the GPU, with every axis but 0 summed (both call sites of the detector) and weights that need
    no gradient: one kernel each way (ops.smooth_l1); else the reference's expression."""
    if bbox_pred.is_cuda and sorted(dim) == list(range(1, bbox_pred.dim())) and not bbox_targets.requires_grad and \
            not bbox_inside_weights.requires_grad and not bbox_outside_weights.requires_grad and \
            bbox_inside_weights.numel() == bbox_outside_weights.numel() and bbox_pred.numel() % max(bbox_inside_weights.numel(), 1) == 0 and \
            bbox_targets.shape == bbox_pred.shape and _trailing_group(bbox_pred, bbox_inside_weights):
        from i2vsgg_amd import ops
        return ops.smooth_l1(bbox_pred, bbox_targets, bbox_inside_weights, bbox_outside_weights, sigma)
    s2 = sigma ** 2
    d = bbox_inside_weights * (bbox_pred - bbox_targets)
    ad = d.abs()
    near = (ad < 1.0 / s2).detach().float()
    loss = bbox_outside_weights * (d * d * (s2 / 2.0) * near + (ad - 0.5 / s2) * (1.0 - near))
    for i in sorted(dim, reverse=True):
        loss = loss.sum(i)
    return loss.mean()


def _trailing_group(pred, w):
    """w broadcasts against pred by repeating each weight over a trailing group: same shape, or same leading axes with the
    remaining ones of length 1."""
    if w.shape == pred.shape:
        return True
    if w.dim() != pred.dim():
        return False
    k = 0
    while k < pred.dim() and w.shape[k] == pred.shape[k]:
        k += 1
    return all(d == 1 for d in w.shape[k:])


def adjust_learning_rate(optimizer, decay=0.1):
    for g in optimizer.param_groups:
        g["lr"] = decay * g["lr"]


def clip_gradient(model, clip_norm):
    sq = [p.grad.norm() ** 2 for p in model.parameters() if p.requires_grad and p.grad is not None]
    total = torch.sqrt(torch.stack(sq).sum()).item() if sq else 0.0
    k = clip_norm / max(total, clip_norm)
    for p in model.parameters():
        if p.requires_grad and p.grad is not None:
            p.grad.mul_(k)


def save_checkpoint(state, filename):
    torch.save(state, filename)


def weights_normal_init(model, dev=0.01):
    for m in (model if isinstance(model, list) else [model]):
        for mod in m.modules():
            w = getattr(mod, "weight", None)
            if isinstance(w, torch.nn.Parameter) and w.dim() >= 2:
                w.data.normal_(0.0, dev)
