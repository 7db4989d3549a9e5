"""``nms(dets, thresh, force_cpu=False)`` with the reference contract (nms/nms_wrapper.py:13-20):
``dets`` (N,5) fp32 [x1,y1,x2,y2,score]; returns ``[]`` for N == 0, else a 1-D int32 CPU tensor of
kept row indices in descending-score order.  The work runs on the GPU (device sort + bitmask NMS
with an on-device suppression scan); the single D2H copy is the returned keep list the reference
API promises.  Hot paths (``_ProposalLayer``) use ``ops.rpn_proposal`` instead and never sync."""
from i2vsgg_amd import ops


def nms(dets, thresh, force_cpu=False):
    if dets.shape[0] == 0:
        return []
    if not dets.is_cuda:
        dets = dets.cuda()
    dets = dets.float()
    order = ops.sort_desc(dets[:, 4].contiguous().view(1, -1))[0].long()
    keep, num = ops.nms_sorted(dets[order].contiguous(), float(thresh))
    k = int(num.item())
    return order[keep[0, :k].long()].int().cpu()


def nms_device(dets, thresh, max_keep=0):
    """Asynchronous variant: returns (kept row indices (n,) int64 on the device, count tensor)."""
    order = ops.sort_desc(dets[:, 4].contiguous().view(1, -1))[0].long()
    keep, num = ops.nms_sorted(dets[order].contiguous(), float(thresh), max_keep)
    return order[keep[0].long().clamp_(0, dets.shape[0] - 1)], num[0]
