"""attention/clustering_feature.py, the offline step that produces the k-means centres the region-attention net
assigns pixels to: per sampled image, the `attention_layer` activation bilinearly up-sampled x2 plus C/16 x-position and
C/16 y-position channels (:373-387) are the points; the reference hands all of them to sklearn's CPU KMeans (:394) and
also carries its own torch `lloyd` (:212-235, forgy init, centre-shift tolerance).

Here the points never leave the GPU and are never materialised as an [N, 576] matrix: Lloyd's two steps run on the
up-sampled activations in place -- `w2e_cluster_assign` (the same kernel the net uses at training time) and
`w2e_cluster_accumulate` (per-cluster sums by a fixed-order reduction, position channels evaluated analytically)."""
import torch
import torch.nn.functional as F

from ._lib import call, ptr, stream_ptr
from .run_attention import _i32ptr, cluster_assign


def clustering_points(blend_feature):
    """:373-381: the x2 bilinear (align_corners=True) up-sampling of the activation; the position channels are implicit
    (2*i/(size-1) - 1 along x and y, C/16 copies each).  Returns [B,C,2s,2s] on the input's device."""
    size = blend_feature.shape[2] * 2
    return F.interpolate(blend_feature.detach(), size=size, mode="bilinear", align_corners=True).contiguous()


def points_matrix(points):
    """The literal [B*s*s, C + 2*(C//16)] matrix of :382-391 (tests / interop with sklearn only)."""
    b, c, s, _ = points.shape
    pc = c // 16
    xs = torch.arange(s, device=points.device).float().unsqueeze(0).repeat(s, 1) * 2 / float(s - 1) - 1
    ys = torch.arange(s, device=points.device).float().unsqueeze(1).repeat(1, s) * 2 / float(s - 1) - 1
    cat = torch.cat([points, xs[None, None].repeat(b, pc, 1, 1), ys[None, None].repeat(b, pc, 1, 1)], 1)
    return cat.permute(0, 2, 3, 1).reshape(-1, c + 2 * pc)


def cluster_sums(points, assign, clusters):
    """(sums [K, D], counts [K]) of the points of every cluster."""
    b, c, s, _ = points.shape
    pc = c // 16
    partial = torch.empty((b, clusters, c + 2 * pc), device=points.device, dtype=torch.float32)
    counts = torch.empty((b, clusters), device=points.device, dtype=torch.float32)
    call("w2e_cluster_accumulate", ptr(points), _i32ptr(assign), ptr(partial), ptr(counts), b, c, pc, s, clusters, stream_ptr())
    return partial.sum(0), counts.sum(0)


def forgy(points, n_clusters, generator=None):
    """:205-209: n_clusters distinct points drawn uniformly."""
    b, c, s, _ = points.shape
    idx = torch.multinomial(torch.ones(b * s * s), n_clusters, generator=generator)
    rows = []
    pc = c // 16
    for i in idx.tolist():
        bi, p = divmod(i, s * s)
        y, x = divmod(p, s)
        pos = torch.tensor([x * 2 / float(s - 1) - 1] * pc + [y * 2 / float(s - 1) - 1] * pc, device=points.device)
        rows.append(torch.cat([points[bi, :, y, x], pos]))
    return torch.stack(rows)


def lloyd(points, n_clusters, tol=1e-4, initial_state=None, max_iter=300, generator=None):
    """:212-235: assign every point to its nearest centre, move the centres to the means, until the summed centre shift,
    squared, falls under `tol`.  points: [B,C,s,s] (clustering_points).  Returns (assign int32 [B,s,s], centres [K,D]).
    A cluster that loses all its points keeps its centre (the reference's mean of an empty selection is NaN)."""
    centres = (initial_state if initial_state is not None else forgy(points, n_clusters, generator)).to(points.device, torch.float32).clone()
    assign = None
    for _ in range(max_iter):
        assign = cluster_assign(points, centres)
        sums, counts = cluster_sums(points, assign, n_clusters)
        new = torch.where(counts[:, None] > 0, sums / counts[:, None].clamp_min(1), centres)
        shift = torch.sum(torch.sqrt(torch.sum((new - centres) ** 2, dim=1)))
        centres = new
        if float(shift) ** 2 < tol:
            break
    return assign, centres
