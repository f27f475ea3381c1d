"""Test double for nets.PartialFC.HipHeadKernels built from the oracle (CPU tensors).  Lets the world_size > 1
host logic of the drop-in PartialFC (all-gather with gradient, label re-basing, sampling + optimizer patching,
the three per-row all-reduces, reduce-scatter x world_size) run on gloo without a GPU.  TEST-ONLY."""
import torch

from oracle import head_ref


class OracleHeadKernels:
    def normalize(self, x):
        xh, n = head_ref.l2_normalize(x.detach())
        return xh, n.reshape(-1)

    @staticmethod
    def _logits(ehat, what, labels, s, m):
        raw = ehat @ what.t()
        z, slope = head_ref.arcface_logits(raw.clamp(-1.0, 1.0), labels.long(), s, m)
        return raw, z, slope

    def forward_stats(self, ehat, what, labels_i32, s, m):
        _, z, _ = self._logits(ehat, what, labels_i32, s, m)
        rmax = z.max(dim=1).values
        rsum = torch.exp(z - rmax[:, None]).sum(dim=1)
        zt = torch.zeros(z.shape[0])
        rows = torch.nonzero(labels_i32 >= 0).flatten()
        zt[rows] = z[rows, labels_i32[rows].long()]
        return zt, rmax, rsum

    def rescale(self, rowsum, local_max, global_max):
        rowsum.mul_(torch.exp(local_max - global_max))

    def target_prob(self, zt, labels_i32, rmax, rsum):
        return torch.where(labels_i32 >= 0, torch.exp(zt - rmax) / rsum, torch.zeros_like(zt))

    def loss(self, q):
        return -(q.clamp_min(1e-30).log().mean()).reshape(1)

    def pack_stats(self, zt, labels_i32, rmax, rsum):
        return torch.stack([rmax, rsum, torch.where(labels_i32 >= 0, zt, torch.full_like(zt, float("-inf")))], dim=1)

    def merge_stats(self, gathered):
        gmax = gathered[:, :, 0].max(dim=0).values
        gsum = (gathered[:, :, 1] * torch.exp(gathered[:, :, 0] - gmax)).sum(dim=0)
        q = torch.exp(gathered[:, :, 2] - gmax).sum(dim=0) / gsum
        return gmax, gsum, q

    def backward(self, ehat, enorm, what, wnorm, labels_i32, s, m, rmax, rsum, n_global, upstream, e_scale=1.0, on_de=None):
        raw, z, slope = self._logits(ehat, what, labels_i32, s, m)
        dz = torch.exp(z - rmax[:, None]) / rsum[:, None]
        rows = torch.nonzero(labels_i32 >= 0).flatten()
        dz[rows, labels_i32[rows].long()] -= 1.0
        dz = dz / n_global * upstream
        dcos = dz * s * slope * ((raw >= -1.0) & (raw <= 1.0))
        d_e = head_ref.l2_normalize_bwd(dcos @ what, ehat, enorm[:, None]) * e_scale
        if on_de is not None:
            on_de(d_e)
        d_w = head_ref.l2_normalize_bwd(dcos.t() @ ehat, what, wnorm[:, None])
        return d_e, d_w

    def gather_rows(self, table, index):
        return table[index].clone()

    def scatter_rows(self, rows, index, table):
        table[index] = rows
