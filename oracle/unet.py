"""CPU oracle: the CQT-octave U-Net denoiser body, restated with plain torch ops (fp32, CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned by tests/golden/unet_*.npz, which were
produced by importing the reference class itself (tests/golden/make_golden.py).

Follows reference networks/unet_cqt_oct_with_projattention_adaLN_2.py:
  * ``embed``            RFF_MLP_Block.forward                  :184-211
  * ``group_std_norm``   BiasFreeGroupNorm.forward              :147-163
  * ``resample_down/up`` UpDownResample.forward                 :549-580  (cubic 8-tap, :514-515)
  * ``time_attention``   TimeAttentionBlock.forward             :338-380
  * ``resnet_block``     ResnetBlock.forward                    :452-493
  * ``OracleUnet.forward`` Unet_CQT_oct_with_attention.forward  :730-845
Parameters are addressed by the reference's state_dict key names, so a reference checkpoint (or the
state_dict of the product module) loads unchanged.
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn.functional as F

CUBIC = (-0.01171875, -0.03515625, 0.11328125, 0.43359375,
         0.43359375, 0.11328125, -0.03515625, -0.01171875)  # unet...py:514-515
RSQRT2 = 1.0 / (2 ** 0.5)


def embed(sd: Dict[str, torch.Tensor], sigma: torch.Tensor) -> torch.Tensor:
    """sigma[B,1] -> emb[B,emb_dim]   (unet...py:184-211)"""
    table = 2 * math.pi * sigma * sd["embedding.RFF_freq"]
    x = torch.cat([torch.sin(table), torch.cos(table)], dim=1)
    for i in range(3):
        x = F.relu(x @ sd[f"embedding.MLP.{i}.weight"].t() + sd[f"embedding.MLP.{i}.bias"])
    return x


def group_std_norm(x: torch.Tensor, gamma: torch.Tensor, groups: int = 8, eps: float = 1e-7) -> torch.Tensor:
    """x / (unbiased std over each of `groups` channel groups + eps) * gamma; the mean is NOT
    subtracted from x (unet...py:147-163)."""
    B, C, Fd, T = x.shape
    xg = x.reshape(B, groups, -1)
    std = xg.std(-1, keepdim=True)
    return (xg / (std + eps)).reshape(B, C, Fd, T) * gamma


def resample_down(x: torch.Tensor) -> torch.Tensor:
    """[..., T] -> [..., T/2]: reflect-pad 3, 8-tap FIR, stride 2 (unet...py:557-558,572)."""
    sh = x.shape
    k = torch.tensor(CUBIC, dtype=x.dtype, device=x.device).view(1, 1, 8)
    xp = F.pad(x.reshape(-1, 1, sh[-1]), (3, 3), mode="reflect")
    return F.conv1d(xp, k, stride=2).reshape(*sh[:-1], -1)


def resample_up(x: torch.Tensor) -> torch.Tensor:
    """[..., T] -> [..., 2T]: reflect-pad 2, transposed 8-tap FIR stride 2, crop 7 (unet...py:559-560,574)."""
    sh = x.shape
    k = torch.tensor(CUBIC, dtype=x.dtype, device=x.device).view(1, 1, 8)
    xp = F.pad(x.reshape(-1, 1, sh[-1]), (2, 2), mode="reflect")
    return F.conv_transpose1d(xp, k, stride=2, padding=7).reshape(*sh[:-1], -1)


def relative_position_bias(weight: torch.Tensor, T: int, max_distance: int = 64) -> torch.Tensor:
    """RelativePositionBias.forward(T, T) (unet...py:266-312): T5-style bucketed bias, weight[num_buckets, H] -> [1,H,T,T]."""
    num_buckets = weight.shape[0] // 2
    q_pos = torch.arange(T, dtype=torch.long)
    rel = q_pos[None, :] - q_pos[:, None]                            # k_pos - q_pos            (:300-302)
    ret = (rel >= 0).to(torch.long) * num_buckets
    n = torch.abs(rel)
    max_exact = num_buckets // 2
    is_small = n < max_exact
    val_if_large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)).long()
    val_if_large = torch.min(val_if_large, torch.full_like(val_if_large, num_buckets - 1))
    bucket = ret + torch.where(is_small, n, val_if_large)
    return weight[bucket].permute(2, 0, 1).unsqueeze(0)             # "m n h -> 1 h m n"     (:309-311)


def time_attention(sd, p: str, x: torch.Tensor, heads: int, rel_pos_max_distance: int = 64) -> torch.Tensor:
    """TimeAttentionBlock.forward (unet...py:338-380).  x[B,C,F,T] -> [B,C,F,T]."""
    B, C, Fd, T = x.shape
    xp = F.conv2d(x, sd[p + "proj_in.weight"])                      # [B,H,F,T]
    xf = xp.reshape(B, heads * Fd, T)
    v = xp.permute(0, 1, 3, 2)                                      # [B,H,T,F]
    qk = F.conv1d(xf, sd[p + "qk.weight"], sd.get(p + "qk.bias"))   # [B,2*H*F,T]   (bias: attention_dict.bias_qkv, :321)
    qk = qk.reshape(B, heads, 2 * Fd, T).permute(0, 1, 3, 2)        # b (h d) t -> b h t d
    q, k = qk[..., :Fd], qk[..., Fd:]
    sim = torch.einsum("bhnd,bhmd->bhnm", q, k)
    if p + "rel_pos.relative_attention_bias.weight" in sd:          # attention_dict.use_rel_pos (:364): added BEFORE the scale
        sim = sim + relative_position_bias(sd[p + "rel_pos.relative_attention_bias.weight"], T, rel_pos_max_distance)
    sim = sim * (float(Fd) ** -0.5)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bhnm,bhmd->bhnd", attn, v).permute(0, 1, 3, 2)  # [B,H,F,T]
    return F.conv2d(out, sd[p + "proj_out.weight"])


def resnet_block(sd, p: str, x_in: torch.Tensor, emb: torch.Tensor, heads: int = 8) -> torch.Tensor:
    """ResnetBlock.forward (unet...py:452-493); structure is read off the state_dict keys."""
    x = x_in
    if p + "proj_in.weight" in sd:
        x = F.conv2d(x, sd[p + "proj_in.weight"])
    if p + "attn_block.qk.weight" in sd:
        g = emb @ sd[p + "affine2.weight"].t() + sd[p + "affine2.bias"]
        s = emb @ sd[p + "gate2.weight"].t() + sd[p + "gate2.bias"]
        h = group_std_norm(x, sd[p + "norm2.gamma"]) * (g[:, :, None, None] + 1)
        h = time_attention(sd, p + "attn_block.", h, heads) * s[:, :, None, None]
        x = (h + x) * RSQRT2
    k = 0
    while p + f"H.{k}.weight" in sd:
        w = sd[p + f"H.{k}.weight"]
        g = emb @ sd[p + f"affine.{k}.weight"].t() + sd[p + f"affine.{k}.bias"]
        s = emb @ sd[p + f"gate.{k}.weight"].t() + sd[p + f"gate.{k}.bias"]
        h = group_std_norm(x, sd[p + f"norm.{k}.gamma"]) * (g[:, :, None, None] + 1)
        dil = (2 ** k, 1)
        h = F.conv2d(F.gelu(h), w, padding="same", dilation=dil if w.shape[-2] > 1 else 1)
        x = (x + h * s[:, :, None, None]) * RSQRT2
        k += 1
    if p + "proj_out.weight" in sd:
        x = F.conv2d(x, sd[p + "proj_out.weight"])
    res = F.conv2d(x_in, sd[p + "res_conv.weight"]) if p + "res_conv.weight" in sd else x_in
    return (x + res) * RSQRT2


class OracleUnet:
    """Functional restatement of Unet_CQT_oct_with_attention (unet...py:583-845)."""

    def __init__(self, num_octs: int, bins_per_oct: int, cqt, heads: int = 8, num_bottleneck_layers: int = 1):
        self.n, self.bpo, self.CQTransform, self.heads = num_octs, bins_per_oct, cqt, heads
        self.nmid = num_bottleneck_layers
        self.sd: Dict[str, torch.Tensor] = {}

    def load_state_dict(self, sd):
        self.sd = {k: v.detach().to(torch.float32).cpu() for k, v in sd.items()}
        return self

    # body on CQT coefficients: list (low octave first) of complex [B,1,bpo,T] -> same structure
    def body(self, X_list: List[torch.Tensor], emb: torch.Tensor) -> List[torch.Tensor]:
        sd, n = self.sd, self.n
        hs = []
        X = pyr = None
        for i in range(n):
            C = torch.view_as_real(X_list[-1 - i].squeeze(1)).permute(0, 3, 1, 2).contiguous()  # [B,2,bpo,T]
            Cin = C
            if f"freq_encodings.{i}.embeddings" in sd:                # use_fencoding (:754-756, AddFreqEncodingRFF.forward :253-263)
                e = sd[f"freq_encodings.{i}.embeddings"]                  # [1, 2N, bpo]
                Cin = torch.cat((C, e[:, :, :, None].expand(C.shape[0], -1, -1, C.shape[-1])), dim=1)
            C2 = resnet_block(sd, f"downs.{i}.0.", Cin, emb, self.heads)
            if i == 0:
                X, pyr = C2, resample_down(C)
            elif i < n - 1:
                pyr = torch.cat((resample_down(C), resample_down(pyr)), dim=2)
                X = torch.cat((C2, X), dim=2)
            else:
                pyr = torch.cat((C, pyr), dim=2)
                X = torch.cat((C2, X), dim=2)
            X = resnet_block(sd, f"downs.{i}.2.", X, emb, self.heads)
            hs.append(X)
            if i < n - 1:
                X = resample_down(X)
            X = (X + F.conv2d(pyr, sd[f"downs.{i}.1.weight"], padding="same")) * RSQRT2
        Xout = None
        for m in range(self.nmid):
            X = resnet_block(sd, f"middle.{m}.1.", X, emb, self.heads)
            Xout = resnet_block(sd, f"middle.{m}.0.", X, emb, self.heads)
        outs = [None] * n
        for i in range(n):
            j = n - 1 - i
            X = torch.cat((X, hs.pop()), dim=1)
            X = resnet_block(sd, f"ups.{i}.1.", X, emb, self.heads)
            Xout = (Xout + resnet_block(sd, f"ups.{i}.0.", X, emb, self.heads)) * RSQRT2
            X = X[:, :, self.bpo:, :]
            Out, Xout = Xout[:, :, : self.bpo, :], Xout[:, :, self.bpo:, :]
            outs[i] = torch.view_as_complex(Out.permute(0, 2, 3, 1).contiguous()).unsqueeze(1)
            if j > 0:
                X, Xout = resample_up(X), resample_up(Xout)
        return outs

    def forward(self, inputs: torch.Tensor, sigma: torch.Tensor) -> torch.Tensor:
        """inputs[B,L], sigma(cnoise)[B,1] -> [B,L]"""
        emb = embed(self.sd, sigma)
        X_list = self.CQTransform.fwd(inputs.unsqueeze(1))
        outs = self.body(X_list, emb)
        pred = self.CQTransform.bwd(outs).squeeze(1)[:, : inputs.shape[-1]]
        return pred

    __call__ = forward
