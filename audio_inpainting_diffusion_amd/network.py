"""CQT-octave U-Net denoiser on MI355X HIP kernels -- drop-in for the reference network class.

Mirrors ``networks.unet_cqt_oct_with_projattention_adaLN_2.Unet_CQT_oct_with_attention`` (reference file
:583-845): same constructor ``(args, device)``, same ``forward(inputs[B,L], sigma[B,1]) -> [B,L]``, same
``.CQTransform`` attribute (``fwd / bwd / apply_hpf_DC``) and the SAME ``state_dict`` keys and shapes, so the
reference's ``setup_network`` (utils/setup.py:46-53), ``load_state_dict`` strategies
(utils/training_utils.py:214-289) and tester run unchanged when ``network.callable`` names this class.

Execution model (MI355X-first, not a translation of the reference's module tree):
  * the whole evaluation is a STATIC LAUNCH PLAN built once per batch size: pre-allocated activation buffers
    (every ResnetBlock step keeps its input, which is exactly what the input-VJP of the guidance branch
    needs), pre-filled C parameter structs, and a flat list of C-ABI calls on torch's current stream --
    no per-call allocation, no host sync, capturable in a HIP graph;
  * concatenations / slices of the reference (:769-774, :814, :821-822) are never materialised: producers
    write into strided views of the consumer's buffer;
  * group-norm + adaLN modulation collapse to one per-(b,c) scale applied inside the conv kernel's LDS
    staging, gate / residual / 1/sqrt(2) live in its epilogue (csrc/aid_conv.hip);
  * all ~194 affine/gate Linears are one stacked matrix evaluated by ``aid_modulation`` once per evaluation.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .cqt import CQTransform

# Winograd forms of a 5x3 layer -> aid_conv2d_params::x_wino / aid_scale_act_params::wino.  4 = F(4,3), 8 = F(8,3) (fused 1-D kernels); the non-fused 2-D forms:
# 45 = F(4,5) x F(4,3) (48 planes), 85 = F(4,5) x F(8,3) (80 planes over groups of eight samples: round 6)
XW_CODE = {0: 0, 4: 1, 8: 2, 45: 3, 85: 4}
W2D_PLANES = {45: 48, 85: 80}
RSQRT2 = 1.0 / math.sqrt(2.0)
SQRT2 = math.sqrt(2.0)
_CUBIC = (-0.01171875, -0.03515625, 0.11328125, 0.43359375, 0.43359375, 0.11328125, -0.03515625, -0.01171875)


# =========================================================================================================
# Parameter containers (names/shapes identical to the reference modules; no forward of their own)
# =========================================================================================================
def _kaiming_uniform(shape, fan_in, scale):
    return (torch.rand(*shape) * 2 - 1) * math.sqrt(3.0 / fan_in) * scale


class _Weight(nn.Module):
    def __init__(self, shape, fan_in, scale, bias_len=0):
        super().__init__()
        self.weight = nn.Parameter(_kaiming_uniform(shape, fan_in, scale))
        if bias_len:
            self.bias = nn.Parameter(torch.zeros(bias_len))


class _Gamma(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(1, n, 1, 1))


class _Kernel(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("kernel", torch.tensor(_CUBIC, dtype=torch.float32))


class _Embedding(nn.Module):
    """RFF_MLP_Block parameters (unet...py:167-182)."""

    def __init__(self, emb_dim, rff_dim=32):
        super().__init__()
        self.RFF_freq = nn.Parameter(16 * torch.randn([1, rff_dim]), requires_grad=False)
        s = math.sqrt(1 / 3)
        self.MLP = nn.ModuleList([_Weight([128, 2 * rff_dim], 2 * rff_dim, s, 128), _Weight([256, 128], 128, s, 256),
                                  _Weight([emb_dim, 256], 256, s, emb_dim)])


class _RelPos(nn.Module):
    """RelativePositionBias parameters (unet...py:266-273): an nn.Embedding(num_buckets, heads) table."""

    def __init__(self, num_buckets, max_distance, heads):
        super().__init__()
        self.num_buckets, self.max_distance = int(num_buckets), int(max_distance)
        self.relative_attention_bias = _Weight([num_buckets, heads], heads, 1.0)

    def buckets(self, T: int) -> torch.Tensor:
        """[T, T] bucket index of (query n, key m), computed with CPU torch ops in the reference's own operation order
        (float32 log, truncation; unet...py:275-305)"""
        nb = self.num_buckets // 2
        pos = torch.arange(T, dtype=torch.long)
        rel = pos[None, :] - pos[:, None]
        ret = (rel >= 0).to(torch.long) * nb
        n = torch.abs(rel)
        max_exact = nb // 2
        val_if_large = max_exact + (torch.log(n.float() / max_exact) / math.log(self.max_distance / max_exact) * (nb - max_exact)).long()
        val_if_large = torch.min(val_if_large, torch.full_like(val_if_large, nb - 1))
        return ret + torch.where(n < max_exact, n, val_if_large)

    def bias_table(self, T: int) -> torch.Tensor:
        """[H, T, T] additive logit bias (RelativePositionBias.forward, unet...py:275-312)"""
        w = self.relative_attention_bias.weight.detach().float()
        return w[self.buckets(T).to(w.device)].permute(2, 0, 1).contiguous()


class _FreqEnc(nn.Module):
    """AddFreqEncodingRFF parameters (unet...py:213-232): a frozen [1, 2N, F] sin/cos table concatenated to an octave's input."""

    def __init__(self, f_dim, N):
        super().__init__()
        self.RFF_freq = nn.Parameter(16 * torch.randn([1, N]), requires_grad=False)
        table = 2 * math.pi * torch.arange(f_dim)[None, None, :] * self.RFF_freq.unsqueeze(-1)
        self.embeddings = nn.Parameter(torch.cat([torch.sin(table), torch.cos(table)], dim=1), requires_grad=False)


class _Attn(nn.Module):
    """TimeAttentionBlock parameters (unet...py:315-336)."""

    def __init__(self, nin, heads, fdim, bias_qkv=False, rel_pos=None):
        super().__init__()
        s = math.sqrt(1 / 3)
        n = heads * fdim
        self.qk = _Weight([2 * n, n, 1], n, s, bias_len=2 * n if bias_qkv else 0)
        self.proj_in = _Weight([heads, nin, 1, 1], nin, s)
        self.proj_out = _Weight([nin, heads, 1, 1], heads, s)
        if rel_pos is not None:
            self.rel_pos = _RelPos(*rel_pos, heads)


class _ResBlock(nn.Module):
    """ResnetBlock parameters (unet...py:383-448)."""

    def __init__(self, dim, dim_out, num_dils, kernel_size, emb_dim, proj_place="before", attention=False, heads=8, fdim=0,
                 bias_qkv=False, rel_pos=None):
        super().__init__()
        self.dim, self.dim_out, self.num_dils, self.ks, self.proj_place = dim, dim_out, num_dils, tuple(kernel_size), proj_place
        self.has_attn, self.heads, self.fdim = attention, heads, fdim
        s, z = math.sqrt(1 / 3), 1e-7
        N = dim_out if proj_place == "before" else dim
        self.N = N
        if proj_place != "before" and N != dim_out:
            self.proj_out = _Weight([dim_out, N, 1, 1], N, s)
        if dim != dim_out:
            self.res_conv = _Weight([dim_out, dim, 1, 1], dim, s)
        if dim != N:
            self.proj_in = _Weight([N, dim, 1, 1], dim, s)
        kh, kw = self.ks
        self.H = nn.ModuleList([_Weight([N, N, kh, kw], N * kh * kw, s) for _ in range(num_dils)])
        self.affine = nn.ModuleList([_Weight([N, emb_dim], emb_dim, s, N) for _ in range(num_dils)])
        self.gate = nn.ModuleList([_Weight([N, emb_dim], emb_dim, z, N) for _ in range(num_dils)])
        self.norm = nn.ModuleList([_Gamma(N) for _ in range(num_dils)])
        if attention:
            self.norm2 = _Gamma(N)
            self.affine2 = _Weight([N, emb_dim], emb_dim, s, N)
            self.gate2 = _Weight([N, emb_dim], emb_dim, z, N)
            self.attn_block = _Attn(N, heads, fdim, bias_qkv, rel_pos)


# =========================================================================================================
# Launch plan
# =========================================================================================================
from .plan import Plan as _Plan   # flat launch list + read / write sets + two-lane schedule (plan.py)


class _Builder:
    def __init__(self, net, B, device, train=False):
        self.net, self.B, self.device = net, B, device
        self.train = bool(train)   # also emit parameter-gradient ops into the backward plan (training step)
        self.S_of = {}             # scale tensor data_ptr -> [B,C] buffer holding sum_{f,t} (dL/d(x*scale) * scale) * x
        self.mod = self.dmod = None   # [B, sum N] modulation vectors and their gradient (set by the network before emission)
        self.pgrad = None          # name -> gradient view (flat gradient buffer laid out like the flat parameter buffer)
        self.params = None         # name -> parameter tensor (its own layout)
        self.plan = _Plan()
        self.nbytes = 0
        self.lane = 0          # lane of the ops emitted next (plan.py): 0 = trunk, 1 = init blocks / pyramid / out blocks
        self.use_lanes = not self.train and bool(getattr(net, "plan_lanes", True))
        self._stats_ws = {}    # lane -> scratch of the two-stage group reductions (lanes run concurrently: one each)
        self._keep = []        # superseded scratch tensors that recorded launches still point into
        self.whole_batch = True   # False for the states of sub-batches (split-K instances are for a whole batch of one only)
        self.bwd = []          # (lane, closure emitting the VJP ops of a forward op); run in reverse by finish_backward
        self._stat_src = {}    # view key of a forward conv output -> (its params struct, partials per (b, group)): see stats()
        self._nb_src = {}      # (reverse sweep) view key of a gradient tensor -> (the aid_norm_bwd params that wrote it last, op index, end address)
        self._g_last = {}      # (reverse sweep) Winograd-domain gradient scratch data_ptr -> index of the last op that used it
        self._in_bwd = False
        self.bplan = None
        self.gmap = {}         # storage data_ptr -> flat gradient storage of the same size
        self.gstate = {}       # storage data_ptr -> 'full' (first contribution overwrites the whole tensor: no zero fill,
                               #   no read-modify-write) | 'zero' (touched through partial views: zero-filled, accumulated)
        self.scratch = {}      # shape -> scratch tensor for dgrad outputs awaiting the normalisation backward
        self.galias = {}       # (reverse sweep) storage data_ptr of a gradient tensor -> [(view key, gradient view, source view, multiplier)]: a PENDING scaled
                               #   copy "gradient = multiplier * source" that has not been launched -- see _defer_copy / Gsrc

    # ---- gradient storage: one flat buffer per activation storage, views share strides/offsets ------------
    def G(self, t, _peek=False):
        st = t.untyped_storage()
        g = self.gmap.get(st.data_ptr())
        if g is None:
            g = self.gmap[st.data_ptr()] = torch.zeros(st.nbytes() // 4, device=self.device, dtype=torch.float32)
            self.nbytes += g.numel() * 4
        gv = g.as_strided(t.size(), t.stride(), t.storage_offset())
        if self.galias and not _peek:                     # anybody who asks for this gradient the ordinary way needs it in memory: launch what was deferred
            self._materialise(g.untyped_storage().data_ptr())      # (galias is keyed by the GRADIENT buffer's storage)
        return gv

    # ---- deferred scaled copies of the reverse sweep (round 6) -------------------------------------------------------------------
    # "dL/dres = c * dL/dy" as the FIRST contribution to dL/dres is a pure scaled copy (8 bytes per element and a launch: 56 of them per guided evaluation,
    # 1.6 % of the GPU time at batch 8 and 61 launches of ~10 us at batch 1).  Where the only readers of dL/dres are the gate pre-pass and the normalisation
    # backward of the LAST dilated step of a block, those read dL/dy through the pending copy instead (aid_scale_act mul, aid_norm_bwd a) and the copy is
    # never launched; any other access (G) launches it first.
    def _defer_copy(self, src, dst, c):
        """dst = c * src, to be launched only if somebody needs dst in memory (both gradient views)"""
        if not self.net.fold_grad_copies or self.train or not self._in_bwd:
            self.add2_raw(src, None, dst, c, 0.0)
            return
        key = dst.untyped_storage().data_ptr()
        self._materialise(key)                             # (an older pending copy into the same storage would be overwritten in the wrong order)
        self.galias.setdefault(key, []).append((self._vkey(dst), dst, src, float(c)))

    def _materialise(self, storage_ptr=None):
        for key in ([storage_ptr] if storage_ptr is not None else list(self.galias)):
            for _vk, dst, src, c in self.galias.pop(key, ()):
                self.add2_raw(src, None, dst, c, 0.0)

    def Gsrc(self, t):
        """(gradient view to READ, multiplier): the source of a pending scaled copy into dL/dt when exactly that view is pending (the copy is then dropped),
        else (G(t), 1.0)"""
        gv = self.G(t, _peek=True)
        key = gv.untyped_storage().data_ptr()
        pend = self.galias.get(key)
        if pend and len(pend) == 1 and pend[0][0] == self._vkey(gv):
            _vk, _dst, src, c = self.galias.pop(key)[0]
            return src, c
        return self.G(t), 1.0

    def _gacc(self, t) -> bool:
        """Decide, at plan-build time, whether the next gradient contribution to activation ``t`` must accumulate
        (True) or may simply overwrite (False: it is the first contribution and covers the whole storage)."""
        st = t.untyped_storage()
        key = st.data_ptr()
        state = self.gstate.get(key)
        if state is None:
            whole = t.is_contiguous() and t.numel() * 4 == st.nbytes()
            self.gstate[key] = "full" if whole else "zero"
            return not whole
        return True

    def _scratch(self, shape):
        key = (self.lane,) + tuple(shape)                 # one set of scratch tensors per lane (they run concurrently)
        t = self.scratch.get(key)
        if t is None:
            t = self.scratch[key] = self.buf(*[d for d in tuple(shape) if not isinstance(d, str)])
        return t

    def _split_ws(self, nbytes):
        """split-K scratch of the row-shared F(4,3) kernel (aid_kernels.h: ws): zero flags + partial accumulators, one per lane"""
        key = (self.lane, "w4r_split")
        t = self.scratch.get(key)
        if t is None or t.numel() * 4 < nbytes:
            if t is not None:
                self._keep.append(t)                      # launches already recorded point into it
            t = self.scratch[key] = torch.zeros(max(nbytes, 4096 + 336 * 98304) // 4, device=self.device, dtype=torch.float32)
            self.nbytes += t.numel() * 4
        if not any(t is z for z in self.plan.zero_on_fail):
            self.plan.zero_on_fail.append(t)               # (its first 4096 bytes are the flags)
        return t

    @property
    def stats_ws(self):
        t = self._stats_ws.get(self.lane)
        if t is None:
            t = self._stats_ws[self.lane] = torch.empty(self.B * 8 * _lib.AID_STATS_SPLIT * 2, device=self.device, dtype=torch.float64)
            self.nbytes += t.numel() * 8
        return t

    def on_lane(self, lane):
        """context manager: ops (and the backward closures registered meanwhile) go to ``lane``"""
        bd = self

        class _L:
            def __enter__(self_):
                self_.prev, bd.lane = bd.lane, (lane if bd.use_lanes else 0)

            def __exit__(self_, *a):
                bd.lane = self_.prev
        return _L()

    def _add(self, name, params, *tensors, **kw):
        self.plan.lane = self.lane
        return self.plan.add(name, params, *tensors, **kw)

    def _reg_bwd(self, fn):
        self.bwd.append((self.lane, fn))

    def finish_backward(self):
        """Emit the reverse sweep (input-VJP) into ``self.bplan``."""
        fwd_plan, self.plan = self.plan, _Plan()
        self._in_bwd = True
        self._stat_src.clear()
        self._nb_src.clear()
        self._g_last.clear()
        for lane, emit in reversed(self.bwd):
            self.lane = lane
            emit()
        self.lane = 0
        self._materialise()                                # (pending copies nobody read through: the caller reads the gradients from memory)
        self._in_bwd = False
        self.bplan, self.plan = self.plan, fwd_plan
        return self.bplan

    def buf(self, *shape):
        t = torch.empty(*shape, device=self.device, dtype=torch.float32)
        self.nbytes += t.numel() * 4
        return t

    # ---- op emitters ---------------------------------------------------------------------------------
    @staticmethod
    def _vkey(t):
        return (t.data_ptr(), tuple(t.shape), tuple(t.stride()))

    def _wrote(self, t):
        """an op (re)writes t: forget what was recorded about overlapping tensors -- forward: the conv whose epilogue could supply the
        statistics of t; reverse sweep: the aid_norm_bwd that could also write the Winograd-domain copy of t"""
        d = self._nb_src if self._in_bwd else self._stat_src
        if not d or t is None:
            return
        lo = t.data_ptr()
        hi = lo + 4 * (1 + sum((n - 1) * st for n, st in zip(t.shape, t.stride())))
        for k in [k for k, v in d.items() if not (v[2] <= lo or hi <= k[0])]:
            del d[k]

    def stats(self, x, gamma, mod, scale, stats=None, gname=None):
        """group statistics of x -> per-(b,c) scale (+ saved mean / inverse std for the VJP).  When x is the untouched output of a forward
        conv on the row-shared F(4,3) kernel, that conv's epilogue writes the (sum, sum of squares) partials and the read pass is skipped."""
        B, Cc, F, T = x.shape
        src = None if self._in_bwd else self._stat_src.pop(self._vkey(x), None)
        ws, ws_n = self.stats_ws, 0
        if src is not None:
            cp, ws_n = src[0], src[1]
            ws = torch.empty(B * 8 * ws_n * 2, device=self.device, dtype=torch.float64)
            self.nbytes += ws.numel() * 8
            cp.stat_ws, cp.stat_n = ws.data_ptr(), ws_n
            src[3].also_writes(ws)                         # (the conv's epilogue now writes the partials this op folds)
            if self.net.fuse_fin and self._fin_wanted(cp.B, cp.x_wino) and _lib.lib().aid_conv2d_fin_supported(cp.B, cp.Cin, cp.Cout, cp.F, cp.T, cp.dilF, cp.x_wino):
                # the last tile of each sample folds the partials itself (aid_kernels.h: fin_mode = 1): no aid_group_stats launch at all
                cnt = self._fin_count(src[3].lane)
                cp.fin_mode, cp.fin_count, cp.fin_gamma, cp.fin_eps = 1, cnt.data_ptr(), gamma.data_ptr(), 1e-7
                cp.fin_mod, cp.fin_mod_ld = _lib.ptr(mod), (0 if mod is None else mod.stride(0))
                cp.fin_scale, cp.fin_stats = scale.data_ptr(), _lib.ptr(stats)
                src[3].also_writes(scale, stats, cnt)
                src[3].also_reads(gamma, mod)
                self.plan.keep.extend(t for t in (ws, gamma, mod, scale, stats, cnt) if t is not None)
                self._stats_train_hook(scale, gamma, mod, gname, stats, B, Cc)
                return
        p = _lib.GroupStatsParams(_lib.view4(x), B, Cc, F, T, 8, gamma.data_ptr(), _lib.ptr(mod),
                                  0 if mod is None else mod.stride(0), 1e-7, scale.data_ptr(), _lib.ptr(stats),
                                  ws.data_ptr(), ws_n)
        self._add("aid_group_stats", p, x, gamma, mod, scale, stats, ws, writes=(scale, stats) + (() if src is not None else (ws,)))
        self._stats_train_hook(scale, gamma, mod, gname, stats, B, Cc)

    @staticmethod
    def _fin_wanted(B, x_wino):
        """The output pass of the 2-D form has thousands of short blocks per sample, each of which would publish its partial past the L2 and bump the
        sample's arrival counter: measured neutral at batch 1 and -0.4 % at batch 8 (profiles/r05_fin2d_ab.txt) -- taken for launches of at most two
        samples, where the fold launch it removes is a larger share of the evaluation; the row-shared kernels (few large tiles) always take it."""
        return x_wino not in (3, 4) or B <= 2

    def _fin_count(self, lane):
        """arrival counters of the fused finalisation (aid_kernels.h: fin_count): one zeroed word per sample and lane (launches of a lane are serial and
        every launch leaves them zero)"""
        key = (lane, "fin_count")
        t = self.scratch.get(key)
        if t is None:
            t = self.scratch[key] = torch.zeros(self.B, device=self.device, dtype=torch.int32)
        if not any(t is z for z in self.plan.zero_on_fail):
            self.plan.zero_on_fail.append(t)               # (a launch that fails must not leave a counter half-way for the next run)
        return t

    def _stats_train_hook(self, scale, gamma, mod, gname, stats, B, Cc):
        if self.train and gname is not None:
            def bw():                                   # gradient of scale = gamma (1 + affine) / (std + eps) w.r.t. gamma and the affine vector
                S = self.S_of.pop(scale.data_ptr(), None)
                if S is None:
                    return
                dm = self.dmod_like(mod)
                sp = _lib.ScaleBwdParams(S.data_ptr(), S.stride(0), scale.data_ptr(), scale.stride(0), gamma.data_ptr(), _lib.ptr(mod),
                                         0 if mod is None else mod.stride(0), stats.data_ptr(), self.pgrad[gname].data_ptr(),
                                         _lib.ptr(dm), 0 if dm is None else dm.stride(0), B, Cc, 8, 1)
                self._add("aid_scale_bwd", sp, S, scale, gamma, mod, stats, dm, writes=(self.pgrad[gname], dm))
            self._reg_bwd(bw)

    def dmod_like(self, view):
        """the slice of the modulation-gradient buffer that corresponds to a slice ``view`` of the modulation buffer"""
        if view is None:
            return None
        off = (view.data_ptr() - self.mod.data_ptr()) // 4
        assert 0 <= off < self.mod.shape[1] and view.stride(0) == self.mod.stride(0)
        return self.dmod[:, off:off + view.shape[1]]

    def _train_conv(self, x, gy, gd, wname, cin, cout, kh, kw, dil, in_scale, act, out_scale, alpha, wpw=None):
        """Parameter-gradient ops of one conv (weight, gate vector, and the <dL/du * scale, x> sums its input scale needs).
        5x3 layers with their F(4,3) weight pack take the Winograd form of the weight gradient (half the MFMAs): both operands are
        transformed first -- gy by aid_wino_gy, the conv input by the aid_scale_act(wino=1) pass that re-creates what the forward saw."""
        B, _, F, T = gy.shape
        K = kh * kw
        wino = bool(self.net.wgrad_wino and (kh, kw) == (5, 3) and T % 16 == 0 and cin >= 32 and cout >= 32
                    and (out_scale is None or (wpw is not None and wpw.shape[0] == 30)))
        xin, isc = x, in_scale
        if wino:
            G6 = 6 * (T // 4)
            xin = self._scratch(("hww", B, cin, F, G6))
            sp = _lib.ScaleActParams(_lib.view4(x), _lib.view4(xin), _lib.ptr(in_scale) if act else None,
                                     in_scale.stride(0) if (act and in_scale is not None) else 0, B, cin, F, T, 1 if act else 0, 1)
            self._add("aid_scale_act", sp, x, xin, in_scale, writes=(xin,))
            if act:
                isc = None
            gyw = self._scratch(("gyw", B, cout, F, G6))
            self._add("aid_wino_gy", _lib.WinoGyParams(_lib.view4(gy), _lib.view4(gyw), B, cout, F, T), gy, gyw, writes=(gyw,))
            gop = gyw
        else:
            gop = gy
            if act:                                       # the conv saw gelu(x * scale): recompute it (one pass) into scratch
                xin = self._scratch(("hw",) + tuple(x.shape))
                sp = _lib.ScaleActParams(_lib.view4(x), _lib.view4(xin), in_scale.data_ptr(), in_scale.stride(0), B, cin, F, T, 1, 0)
                self._add("aid_scale_act", sp, x, xin, in_scale, writes=(xin,))
                isc = None
        tiles = int(_lib.lib().aid_conv2d_wgrad_tiles(cin, cout, kh, kw, int(wino)))
        S = max(1, min(F, 256 // (tiles * B)))            # about one workgroup per CU: every extra split is another partial to write and reduce
                                                          # (1024 / 768 / 512 / 256 workgroups: 250 / 238 / 227 / 221 ms per iteration at batch 4)
        KP = 30 if wino else K                            # taps per (co, ci) in the partials (U domain: xi * 5 + kh)
        P = self._scratch(("P", B * S * cout * cin * KP))
        wp = _lib.WgradParams(_lib.view4(gop), _lib.view4(xin), P.data_ptr(), B, cin, cout, F, T, kh, kw, dil, S, alpha, int(wino))
        self._add("aid_conv2d_wgrad", wp, gop, xin, P, flops=2 * B * F * T * cin * cout * K, writes=(P,))
        W = self.params[wname]
        dg = self.dmod_like(out_scale)
        rp = _lib.WgradReduceParams(P.data_ptr(), W.data_ptr(), _lib.ptr(out_scale), 0 if out_scale is None else out_scale.stride(0),
                                    _lib.ptr(isc), 0 if isc is None else isc.stride(0), self.pgrad[wname].data_ptr(), _lib.ptr(dg),
                                    0 if dg is None else dg.stride(0), B, S, cout, cin, K, 1, int(wino),
                                    _lib.ptr(wpw) if wino else None, wpw.shape[1] if (wino and wpw is not None) else 0,
                                    wpw.shape[2] if (wino and wpw is not None) else 0)
        self._add("aid_wgrad_reduce", rp, P, W, out_scale, isc, dg, wpw, writes=(self.pgrad[wname], dg))
        if in_scale is not None and gd is not None:       # gd = dL/d(x*scale) * scale  ->  S[b,c] = sum gd * x  (aid_scale_bwd divides by scale)
            Sb = self.buf(B, cin)
            cp = _lib.ChannelDotParams(_lib.view4(gd), _lib.view4(x), Sb.data_ptr(), Sb.stride(0), B, cin, F, T)
            self._add("aid_channel_dot", cp, gd, x, Sb, writes=(Sb,))
            self.S_of[in_scale.data_ptr()] = Sb

    def _conv_raw(self, x, y, wp, cin, cout, kh, kw, dil, in_scale, act, out_scale, res, res_scale, alpha, epi=0,
                  aux=None, aux_scale=None, wpw=None, x_wino=0, dot=None, x2=None, fin_stats=None):
        """``x_wino``: 0 plain activations, 4 / 8: ``x`` is the F(4,3) / F(8,3) input transform and ``wpw`` the matching 30- / 50-tap pack."""
        B, _, F, T = y.shape
        x_wino = int(x_wino)
        xw_code = XW_CODE[x_wino]                            # aid_conv2d_params::x_wino
        p = _lib.Conv2dParams()
        nxi = W2D_PLANES.get(x_wino, 0)
        if nxi:                                              # 2-D form: x is the flat V [48 | 80][cin][N] of aid_scale_act(wino = 3 | 4)
            npos = self._w2d_positions(x_wino, B, F, T, dil)
            assert x.dim() == 1 and x.numel() == nxi * cin * npos and x2 is None and in_scale is None and not act
            p.x = _lib.View(x.data_ptr(), 0, 0, 0)
        else:
            assert x.shape[1] + (0 if x2 is None else x2.shape[1]) == cin and y.shape[1] == cout and x.shape[0] == B and x.shape[2] == F
            assert x.shape[3] == {0: T, 4: 6 * (T // 4), 8: 10 * (T // 8)}[x_wino]
            p.x = _lib.view4(x)
        p.y, p.res, p.aux = _lib.view4(y), _lib.view4(res), _lib.view4(aux)
        p.wp = wp.data_ptr()
        p.in_scale, p.in_scale_ld = _lib.ptr(in_scale), (0 if in_scale is None else in_scale.stride(0))
        p.out_scale, p.out_scale_ld = _lib.ptr(out_scale), (0 if out_scale is None else out_scale.stride(0))
        p.aux_scale, p.aux_scale_ld = _lib.ptr(aux_scale), (0 if aux_scale is None else aux_scale.stride(0))
        p.B, p.Cin, p.Cout, p.F, p.T = B, cin, cout, F, T
        p.Cin_pad, p.Cout_pad = wp.shape[1], wp.shape[2]
        p.KH, p.KW, p.dilF, p.act, p.epi = kh, kw, dil, act, epi
        p.alpha, p.res_scale = alpha, res_scale
        p.wp_wino = _lib.ptr(wpw)
        p.wino_taps = 0 if wpw is None else wpw.shape[0]
        p.x_wino = xw_code
        if dot is not None:                              # (buffer, partials per (b, group)): <y, aux> folded into the epilogue
            p.dot_ws, p.dot_n = dot[0].data_ptr(), dot[1]
        cnt = None
        if fin_stats is not None:                        # fin_mode = 2: coefficients of aid_norm_bwd into the floats behind the partials (see _dot_ws)
            cnt = self._fin_count(self.lane)
            p.fin_mode, p.fin_count, p.fin_eps, p.fin_stats = 2, cnt.data_ptr(), 1e-7, fin_stats.data_ptr()
            p.fin_scale = dot[0].data_ptr() + 8 * B * 8 * dot[1]
        if x2 is not None:                               # K axis in two tensors (aid_kernels.h: x2 / Cin1)
            p.x2, p.Cin1 = _lib.view4(x2), x.shape[1]
        ws = None
        if F == 1 and kh == 1 and epi == 0:              # qk projections: few columns, long K -> split-K scratch (aid_kernels.h)
            ws = self._scratch(("ws", 8 * B * cout * T))
            p.ws, p.ws_bytes = ws.data_ptr(), ws.numel() * 4
        elif nxi:                                        # the GEMM's output M [48 | 80][cout][N], read back by the output-transform pass of the same call
            ws = self._scratch(("m45", nxi * cout * npos))
            p.ws, p.ws_bytes = ws.data_ptr(), ws.numel() * 4
        elif x_wino == 4 and B == 1 and self.whole_batch:     # a WHOLE batch of one (never a sub-batch: a segment's bits must not depend on the split):
                                                         # launches with few tiles share the K axis of a tile between two workgroups
            need = int(_lib.lib().aid_conv2d_wino_split_ws_bytes(B, cin, cout, F, T, dil))
            if need:
                ws = self._split_ws(need)
                p.ws, p.ws_bytes = ws.data_ptr(), ws.numel() * 4
        assert wp.shape[0] == kh * kw and (wpw is None or (wpw.shape[0] == {8: 50, 45: 48, 85: 80}.get(x_wino, 30) and wpw.shape[1:] == wp.shape[1:]))
        # algorithmic HBM bytes: x once, residual / aux once, y once, weights once
        nb = 4 * (B * F * T * (cin + cout * (1 + (res is not None) + (aux is not None))) + cin * cout * kh * kw)
        dws = None if dot is None else dot[0]
        if nxi:
            # (bookkeeping only: the launches whose GEMM folds the row-axis output transform write and re-read HALF of M -- the rule of w2d_fold_m in
            #  csrc/aid_wino2d.hip, restated here for the algorithmic byte counts bench.py reports; the library decides on its own)
            mxi = nxi // 2 if (x_wino == 85 and cin <= 128 and 10 * (-(-npos // 64)) * (-(-cout // 128)) >= 768) else nxi
            # two plan nodes on one parameter block: the MFMA-bound batched GEMM M = U V (its FLOPs and the bytes of V + M are booked here) and the
            # HBM-bound output-transform pass with the epilogue (reads M, residual / aux; writes y and the partials)
            self._add("aid_conv2d_wino2d_gemm", p, x, wpw, ws, flops=2 * B * F * T * cin * cout * kh * kw, nbytes=4 * npos * (nxi * cin + mxi * cout), writes=(ws,))
            op = self._add("aid_conv2d_wino2d_output", p, ws, y, res, out_scale, aux, aux_scale, dws, fin_stats, cnt,
                           nbytes=4 * (mxi * npos * cout + B * F * T * cout * (1 + (res is not None) + (aux is not None))), writes=(y, dws, cnt))
        else:
            op = self._add("aid_conv2d", p, x, x2, y, res, wp, in_scale, out_scale, aux, aux_scale, wpw, ws, dws, fin_stats, cnt, flops=2 * B * F * T * cin * cout * kh * kw,
                           nbytes=nb, writes=(y, ws, dws, cnt))
        self._wrote(y)
        if not self._in_bwd and x_wino and epi == 0 and dot is None and self.net.epilogue_stats:
            n = int(_lib.lib().aid_conv2d_stat_partials(B, cin, cout, F, T, dil, xw_code))
            if n:                                         # a later stats(y) may ask this conv's epilogue for the partial sums
                hi = y.data_ptr() + 4 * (1 + sum((m - 1) * st for m, st in zip(y.shape, y.stride())))
                self._stat_src[self._vkey(y)] = (p, n, hi, op)

    def _dot_ws(self, n):
        """scratch for the per-tile <gd, x> partials written by the conv epilogue: [B*8, n] doubles + B*8 floats (coef)"""
        key = (self.lane, "dot", n)
        t = self.scratch.get(key)
        if t is None:
            t = self.scratch[key] = torch.zeros(self.B * 8 * (n + 1), device=self.device, dtype=torch.float64)
            self.nbytes += t.numel() * 8
        return t

    @staticmethod
    def _w2d_positions(form, B, F, T, dil):
        """positions per transform plane of a 2-D-form launch: groups of 4 samples (form 45) or 8 (form 85)"""
        n = int(_lib.lib().aid_conv2d_wino2d_positions(B, F, T, dil))
        return n // 2 if form == 85 else n

    def _wino_input(self, cin, cout, T, wp, wpw, F=0, dil=1, wpw8=None, wpw2=None, wpw3=None):
        """The Winograd form the pre-pass should write for this 5x3 layer (aid_scale_act wino = 1 / 2 -> aid_conv2d x_wino = 1 / 2): 8 = F(8,3)
        (10 MFMAs per 8 outputs; needs the 50-tap pack), 4 = F(4,3), 0 = plain activations.  The library answers from the launch shape
        (aid_conv2d_wino_form); ``net.wino_forms`` restricts the choice (A/B measurements, tests of the F(4,3) kernels)."""
        if wpw is None or wpw.shape[0] != 30:
            return 0
        fb = int(self.net.form_batch or self.B)               # the batch the FORM is chosen for (network.form_batch; default: this launch's)
        if wpw2 is not None and (45 in self.net.wino_forms or 85 in self.net.wino_forms) and _lib.lib().aid_conv2d_wino2d_supported(cin, cout, F, T, dil):
            # the non-fused 2-D form F(4,5) x F(4,3) (csrc/aid_wino2d.hip: 3.0 products per output) where the library predicts it faster than the fused
            # 1-D kernels (aid_conv2d_wino2d_wanted, a function of the launch shape); wino_forms = (45,): wherever it is supported (tests, A/B)
            if tuple(self.net.wino_forms) in ((45,), (85,), (45, 85)) or (min(cin, cout) >= self.net.w2d_min_channels and (T <= self.net.w2d_force_max_T or _lib.lib().aid_conv2d_wino2d_wanted(fb, cin, cout, F, T, dil))
                                                                            and (cout % 128 == 0 or T <= self.net.w2d_c96_max_T)):
                # ... and which T form: F(8,3) along T (85: 2.5 products per output, 2.5 x the activation in V / M) where the library says so
                # (aid_conv2d_wino2d_tform), F(4,3) (45) otherwise; wino_forms without 85 / 45 restricts the choice
                if wpw3 is not None and 85 in self.net.wino_forms and T % 32 == 0 and (45 not in self.net.wino_forms or tuple(self.net.wino_forms) == (85,) or cout % 128 != 0
                                                                                     or int(_lib.lib().aid_conv2d_wino2d_tform(fb, cin, cout, F, T, dil)) == 8):
                    return 85
                if 45 in self.net.wino_forms:
                    return 45
        form = int(_lib.lib().aid_conv2d_wino_form(fb, cin, cout, F, T, dil))
        if form == 8 and (wpw8 is None or 8 not in self.net.wino_forms):
            form = 4
        elif form == 4 and 4 not in self.net.wino_forms and wpw8 is not None and _lib.lib().aid_conv2d_wino8_supported(cin, cout, F, T, dil):
            form = 8                                          # wino_forms = (8,): F(8,3) wherever its tiles fit (tests, A/B)
        if form == 4 and not ((4 in self.net.wino_forms or tuple(self.net.wino_forms) in ((45,), (85,), (45, 85))) and bool(_lib.lib().aid_conv2d_wino_input_ok(self.B, cin, cout, F, T, dil))):
            form = 0
        return form

    @staticmethod
    def _wino_cols(form, T):
        return {0: T, 4: 6 * (T // 4), 8: 10 * (T // 8)}[form]

    def _wino_scratch(self, tag, form, B, C, F, T, dil):
        """scratch for the conv input the pre-pass writes: [B, C, F, cols] (plain / 1-D Winograd domain) or the flat V [48][C][N] of the 2-D form"""
        if form in W2D_PLANES:
            return self._scratch((tag + "45", W2D_PLANES[form] * C * self._w2d_positions(form, B, F, T, dil)))
        return self._scratch((tag, B, C, F, self._wino_cols(form, T)))

    @staticmethod
    def _sa_out(t):
        """aid_scale_act's output view: 4-D scratch, or the flat V of the 2-D form (strides unused)"""
        return _lib.View(t.data_ptr(), 0, 0, 0) if t.dim() == 1 else _lib.view4(t)

    def conv(self, x, y, wp, cin, cout, kh=1, kw=1, dil=1, in_scale=None, act=0, out_scale=None, res=None,
             res_scale=1.0, alpha=1.0, wpT=None, norm_stats=None, wpw=None, wpwT=None, res_nograd=False, wname=None, wpw8=None, wpw8T=None,
             wpw2=None, wpw2T=None, wpw3=None, wpw3T=None):
        """Forward conv + registration of its input-VJP.  ``norm_stats``: the (mean, 1/(std+eps)) buffer when
        ``in_scale`` was produced by ``stats`` from this same ``x`` (the scale then depends on x)."""
        if act and kh > 1:
            # evaluate norm*mod -> GELU once per element into a scratch tensor; the conv stages plain copies
            xw = self._wino_input(cin, cout, x.shape[3], wp, wpw, x.shape[2], dil, wpw8, wpw2, wpw3)
            hbuf = self._wino_scratch("h", xw, x.shape[0], cin, x.shape[2], x.shape[3], dil)
            sp = _lib.ScaleActParams(_lib.view4(x), self._sa_out(hbuf), in_scale.data_ptr(), in_scale.stride(0), x.shape[0], cin,
                                     x.shape[2], x.shape[3], 1, XW_CODE[xw], dil)
            self._add("aid_scale_act", sp, x, hbuf, in_scale, writes=(hbuf,))
            self._conv_raw(hbuf, y, wp, cin, cout, kh, kw, dil, None, 0, out_scale, res, res_scale, alpha, wpw={8: wpw8, 45: wpw2, 85: wpw3}.get(xw, wpw), x_wino=xw)
        else:
            self._conv_raw(x, y, wp, cin, cout, kh, kw, dil, in_scale, act, out_scale, res, res_scale, alpha)
        if wpT is None:
            return

        def bw():
            fused_res = (norm_stats is not None) and (res is x)
            # the last dilated step of a block may read dL/dy THROUGH a pending scaled copy (see _defer_copy): its only readers here are the gate pre-pass
            # (aid_scale_act mul) and the normalisation backward's skip term (a)
            gy, gmul = self.Gsrc(y) if (kh > 1 and out_scale is not None and norm_stats is not None and fused_res and not self.train) else (self.G(y), 1.0)
            B, _, F, T = x.shape
            if res is not None and not fused_res and not res_nograd:
                gr = self.G(res)
                if self._gacc(res):
                    self.add2_raw(gr, gy, gr, 1.0, alpha * res_scale)
                else:
                    self.add2_raw(gy, None, gr, alpha * res_scale, 0.0)
            gin, gsc, gw = gy, out_scale, 0
            if kh > 1 and out_scale is not None:
                # 5x3 dgrad: apply the gate in a copy pass so the conv input needs no in-kernel prologue
                # (keeps it on the direct-to-LDS kernel)
                gw = self._wino_input(cout, cin, gy.shape[3], wpT, wpwT, gy.shape[2], dil, wpw8T, wpw2T, wpw3T) if norm_stats is not None else 0
                gin = self._wino_scratch("g", gw, gy.shape[0], cout, gy.shape[2], gy.shape[3], dil)
                nb = self._nb_src.pop(self._vkey(gy), None) if (gw in (4, 8, 45, 85) and self.net.fuse_norm_bwd_wino and gmul == 1.0) else None
                if nb is not None and nb[1] > self._g_last.get(gin.data_ptr(), -1) and nb[3].lane == self.lane:
                    # gy was written last by an aid_norm_bwd and nothing used this scratch since: that pass also writes gin (for the 2-D forms it then
                    # IS this layer's input pass: the normalisation backward folded into the transform, one read of gd / x / gy instead of two passes)
                    nb[0].wout, nb[0].wscale, nb[0].wscale_ld = self._sa_out(gin), out_scale.data_ptr(), out_scale.stride(0)
                    nb[0].wform, nb[0].wdil = {4: 1, 8: 2, 45: 3, 85: 4}[gw], dil
                    self.plan.keep.extend((gin, out_scale))
                    nb[3].also_writes(gin)
                else:
                    sp = _lib.ScaleActParams(_lib.view4(gy), self._sa_out(gin), out_scale.data_ptr(), out_scale.stride(0), gy.shape[0],
                                             cout, gy.shape[2], gy.shape[3], 0, XW_CODE[gw], dil, 0.0 if gmul == 1.0 else gmul)
                    self._add("aid_scale_act", sp, gy, gin, out_scale, writes=(gin,))
                self._g_last[gin.data_ptr()] = len(self.plan.ops)      # (the dgrad conv below reads it)
                gsc = None
            if norm_stats is not None:
                gd = self._scratch(x.shape)
                # <gd, x> per (sample, group): folded into the dgrad conv's epilogue when it runs on the F(4,3) kernels
                nd = 0
                if act and kh == 5 and gsc is None and wpwT is not None and wpwT.shape[0] == 30:
                    nd = int(_lib.lib().aid_conv2d_dot_partials(B, cout, cin, F, T, dil, XW_CODE[gw]))
                elif act and kh == 1 and kw == 1 and self.net.fuse_dot_1x1:        # 1x1 steps (init / out blocks): the direct-to-LDS kernel's epilogue
                    nd = int(_lib.lib().aid_conv2d_dot_partials_1x1(B, cout, cin, F, T))
                dws = self._dot_ws(nd) if nd else self.stats_ws
                # the last tile of each sample also folds the partials into the normalisation-backward coefficients (aid_kernels.h: fin_mode = 2)
                fin = bool(nd and kh == 5 and gw in (4, 8, 45, 85) and self.net.fuse_fin and self._fin_wanted(B, XW_CODE[gw])
                           and _lib.lib().aid_conv2d_fin_supported(B, cout, cin, F, T, dil, XW_CODE[gw]))
                self._conv_raw(gin, gd, wpT, cout, cin, kh, kw, dil, gsc, 0, in_scale, None, 1.0, alpha,
                               epi=1 if act else 0, aux=x if act else None, aux_scale=in_scale if act else None, wpw={8: wpw8T, 45: wpw2T, 85: wpw3T}.get(gw, wpwT), x_wino=gw,
                               dot=(dws, nd) if nd else None, fin_stats=norm_stats if fin else None)
                if not nd:
                    dp = _lib.GroupDotParams(_lib.view4(gd), _lib.view4(x), B, cin, F, T, 8, self.stats_ws.data_ptr())
                    self._add("aid_group_dot", dp, gd, x, self.stats_ws, writes=(self.stats_ws,))
                if self.train and wname is not None:
                    self._train_conv(x, gy, gd, wname, cin, cout, kh, kw, dil, in_scale, act, out_scale, alpha, wpw=wpw)
                npar = _lib.NormBwdParams(_lib.view4(gd), _lib.view4(x), _lib.view4(gy if fused_res else None), _lib.view4(self.G(x)),
                                          B, cin, F, T, 8, norm_stats.data_ptr(), dws.data_ptr(), 1e-7,
                                          alpha * res_scale * gmul, 1 if self._gacc(x) else 0, nd)
                npar.coef_ready = 1 if fin else 0
                gxv = self.G(x)
                nop = self._add("aid_norm_bwd", npar, gd, x, gy, norm_stats, dws, gxv, writes=(gxv,))
                self._wrote(gxv)
                if npar.accumulate == 0 and x.shape[3] % 16 == 0:
                    hi = gxv.data_ptr() + 4 * (1 + sum((m - 1) * st for m, st in zip(gxv.shape, gxv.stride())))
                    self._nb_src[self._vkey(gxv)] = (npar, len(self.plan.ops) - 1, hi, nop)
            else:
                gx = self.G(x)
                self._conv_raw(gin, gx, wpT, cout, cin, kh, kw, dil, gsc, 0, in_scale, gx if self._gacc(x) else None, 1.0 / alpha, alpha,
                               epi=1 if act else 0, aux=x if act else None, aux_scale=in_scale if act else None)
                if self.train and wname is not None:
                    assert in_scale is None, "a scaled conv input without its statistics buffer has no parameter-gradient path"
                    self._train_conv(x, gy, None, wname, cin, cout, kh, kw, dil, None, act, out_scale, alpha, wpw=wpw)
        self._reg_bwd(bw)


    def pair_dgrad(self, xin, x0, yout, wpair, n, cout):
        """Register the MERGED input gradient of a ResnetBlock's proj_in (xin -> x0) and res_conv (xin -> yout) at the place of proj_in
        (so that it runs after the block's steps in the reverse sweep): dL/dxin (+)= [W_pi^T ; alpha W_rc^T] . [dL/dx0 ; dL/dyout] as ONE 1x1
        conv whose K axis lives in two tensors, instead of two read-modify-write passes over dL/dxin."""
        def bw():
            g1, g2, gx = self.G(x0), self.G(yout), self.G(xin)
            self._conv_raw(g1, gx, wpair, 2 * n, cout, 1, 1, 1, None, 0, None, gx if self._gacc(xin) else None, 1.0, 1.0, x2=g2)
        self._reg_bwd(bw)

    def res_grad(self, res, y, c):
        """Register dL/dres (+)= c * dL/dy (the residual input of a conv whose own input gradient is taken elsewhere)."""
        def bw():
            gy = self.G(y)
            if self._gacc(res):
                gr = self.G(res)
                self.add2_raw(gr, gy, gr, 1.0, c)
            else:
                self._defer_copy(gy, self.G(res, _peek=True), c)
        self._reg_bwd(bw)

    def add2_raw(self, u, v, y, a, b):
        """y = a*u + b*v   (v may be None: y = a*u)"""
        B, Cc, F, T = u.shape
        p = _lib.Add2Params(_lib.view4(u), _lib.view4(v), _lib.view4(y), B, Cc, F, T, a, b)
        self._add("aid_add2", p, u, v, y, writes=(y,))
        self._wrote(y)

    def add2(self, u, v, y, a, b):
        self.add2_raw(u, v, y, a, b)

        def bw():
            gy = self.G(y)
            for tt, c in ((u, a), (v, b)):
                if self._gacc(tt):
                    gt = self.G(tt)
                    self.add2_raw(gt, gy, gt, 1.0, c)
                else:
                    self._defer_copy(gy, self.G(tt, _peek=True), c)
        self._reg_bwd(bw)

    def copy(self, u, y):
        """y = u (strided views), with its VJP"""
        self.add2_raw(u, None, y, 1.0, 0.0)

        def bw():
            gy, gu = self.G(y), self.G(u)
            if self._gacc(u):
                self.add2_raw(gu, gy, gu, 1.0, 1.0)
            else:
                self.add2_raw(gy, None, gu, 1.0, 0.0)
        self._reg_bwd(bw)

    def _resample_raw(self, x, y, up, adjoint=0, accumulate=0):
        B, Cc, F, T = x.shape
        p = _lib.ResampleParams(_lib.view4(x), _lib.view4(y), B, Cc, F, T, int(up), int(adjoint), int(accumulate))
        self._add("aid_resample", p, x, y, writes=(y,))
        self._wrote(y)

    def resample(self, x, y, up):
        self._resample_raw(x, y, up)
        self._reg_bwd(lambda: self._resample_raw(self.G(y), self.G(x), up, adjoint=1, accumulate=1 if self._gacc(x) else 0))

    def attention(self, qk, v, out, heads, F, T, bias=None, relpos=None):
        """``relpos`` (training, use_rel_pos): (bucket index tensor [T, T] int32, parameter name of the [num_buckets, heads] embedding)"""
        B = v.shape[0]
        probs = self.buf(B, heads, T, T)
        scale = float(F) ** -0.5
        p = _lib.AttentionParams(qk.data_ptr(), v.data_ptr(), out.data_ptr(), probs.data_ptr(), B, heads, F, T, scale, _lib.ptr(bias))
        self._add("aid_time_attention", p, qk, v, out, probs, bias, flops=4 * B * heads * T * T * F, writes=(out, probs))
        self._wrote(out)

        def bw():
            gq, gv, go = self.G(qk), self.G(v), self.G(out)
            assert gq.is_contiguous() and gv.is_contiguous() and go.is_contiguous()
            self._gacc(qk)                                   # d(qk) is written, never accumulated
            dsws = self._scratch(("ds", B, heads, T, T))
            bp = _lib.AttentionBwdParams(qk.data_ptr(), v.data_ptr(), probs.data_ptr(), go.data_ptr(), gq.data_ptr(), gv.data_ptr(),
                                         B, heads, F, T, scale, 1 if self._gacc(v) else 0, dsws.data_ptr())
            self._add("aid_time_attention_bwd", bp, qk, v, probs, go, gq, gv, dsws, flops=10 * B * heads * T * T * F, writes=(gq, gv, dsws))
            self._wrote(gq)
            self._wrote(gv)
            if self.train and relpos is not None:           # d(bias table) = dS summed over samples, scattered back onto the embedding rows
                bucket, wname = relpos
                dW = self.pgrad[wname]
                rp = _lib.RelposBwdParams(dsws.data_ptr(), bucket.data_ptr(), dW.data_ptr(), B, heads, T, dW.shape[0], 1)
                self._add("aid_relpos_bwd", rp, dsws, bucket, dW, writes=(dW,))
        self._reg_bwd(bw)

    def bias_grad(self, y, wname):
        """training: gradient of a per-channel bias that was broadcast over (b, f, t) onto ``y`` (bias_qkv: the qk conv's residual input)"""
        def bw():
            gy = self.G(y)
            B, Cc, F, T = gy.shape
            ones = self.scratch.get(("ones", T))
            if ones is None:
                ones = self.scratch[("ones", T)] = torch.ones(max(T, 4), device=self.device, dtype=torch.float32)
            S = self.buf(B, Cc)
            ov = _lib.View(ones.data_ptr(), 0, 0, 0)
            cp = _lib.ChannelDotParams(_lib.view4(gy), ov, S.data_ptr(), S.stride(0), B, Cc, F, T)
            self._add("aid_channel_dot", cp, gy, ones, S, writes=(S,))
            dW = self.pgrad[wname]
            rp = _lib.WgradReduceParams(S.data_ptr(), self.params[wname].data_ptr(), None, 0, None, 0, dW.data_ptr(), None, 0,
                                        B, 1, Cc, 1, 1, 1, 0, None, 0, 0)
            self._add("aid_wgrad_reduce", rp, S, dW, writes=(dW,))
        self._reg_bwd(bw)


# =========================================================================================================
class Unet_CQT_oct_with_attention(nn.Module):
    """Drop-in replacement for the reference class of the same name (unet...py:583)."""

    def __init__(self, args, device):
        super().__init__()
        self.args = args
        net = args.network if hasattr(args, "network") else args
        self.depth = self.num_octs = int(net.cqt.num_octs)
        self.bins_per_oct = int(net.cqt.bins_per_oct)
        self.emb_dim = int(net.emb_dim)
        has_attn = any(int(v) for v in net.attention_layers)
        if not net.use_norm:
            raise NotImplementedError("use_norm=False is not built (every shipped configuration normalises)")
        if net.bottleneck_type != "res_dil_convs":
            raise NotImplementedError("bottleneck type not implemented")   # same error as unet...py:694
        self.heads = int(net.attention_dict.num_heads)
        self.Ns = [int(v) for v in net.Ns]
        self.num_dils = [int(v) for v in net.num_dils]
        self.attention_layers = [int(v) for v in net.attention_layers]
        self.num_bottleneck_layers = int(net.num_bottleneck_layers)
        self.device = torch.device(device)
        win = ("kaiser", float(net.cqt.beta)) if net.cqt.window == "kaiser" else net.cqt.window
        self.CQTransform = CQTransform(self.num_octs, self.bins_per_oct, mode="oct", window=win,
                                       fs=args.exp.sample_rate, audio_len=args.exp.audio_len, dtype=torch.float32,
                                       device=self.device, rules=net.cqt.get("rules") if hasattr(net.cqt, "get") else None)
        n, E, bpo, H = self.num_octs, self.emb_dim, self.bins_per_oct, self.heads
        self.embedding = _Embedding(E)
        ad = net.attention_dict
        akw = dict(heads=H, bias_qkv=bool(ad.bias_qkv),
                   rel_pos=(int(ad.rel_pos_num_buckets), int(ad.rel_pos_max_distance)) if ad.use_rel_pos else None)
        self.use_fencoding = bool(net.use_fencoding)
        self.n_fenc = 32                                 # N_freq_encoding (unet...py:627)
        if self.use_fencoding:                           # registered between `embedding` and the resamplers, as in the reference (:629)
            self.freq_encodings = nn.ModuleList([_FreqEnc(bpo, self.n_fenc) for _ in range(n)])
        Nin = 2 * self.n_fenc + 2 if self.use_fencoding else 2
        self.Nin = Nin
        self.downsamplerT, self.upsamplerT = _Kernel(), _Kernel()
        self.downs, self.middle, self.ups = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        s = math.sqrt(1 / 3)
        for i in range(n):
            dim_in = self.Ns[i] if i == 0 else self.Ns[i - 1]
            dim_out = self.Ns[i]
            self.downs.append(nn.ModuleList([
                _ResBlock(Nin, dim_in, 1, (1, 1), E),
                _Weight([dim_out, 2, 5, 3], 2 * 15, s),
                _ResBlock(dim_in, dim_out, self.num_dils[i], (5, 3), E, attention=bool(self.attention_layers[i]),
                          fdim=(i + 1) * bpo, **akw)]))
        for _ in range(self.num_bottleneck_layers):
            self.middle.append(nn.ModuleList([
                _ResBlock(self.Ns[-1], 2, 1, (1, 1), E, proj_place="after"),
                _ResBlock(self.Ns[-1], self.Ns[-1], self.num_dils[-1], (5, 3), E, attention=bool(self.attention_layers[-1]),
                          fdim=n * bpo, **akw)]))
        for i in range(n - 1, -1, -1):
            dim_in = self.Ns[i] * 2
            dim_out = self.Ns[i] if i == 0 else self.Ns[i - 1]
            self.ups.append(nn.ModuleList([
                _ResBlock(dim_out, 2, 1, (1, 1), E, proj_place="after"),
                _ResBlock(dim_in, dim_out, self.num_dils[i], (5, 3), E, attention=bool(self.attention_layers[i]),
                          fdim=(i + 1) * bpo, **akw)]))
        self._packed: Dict[str, torch.Tensor] = {}
        self._packed_ver = None
        self._states: Dict[tuple, dict] = {}
        self._mod_layout = None
        self.to(self.device)

    # ---------------------------------------------------------------------------------------------------
    # weight packing (kernel-side layouts; refreshed in place when parameters change)
    # ---------------------------------------------------------------------------------------------------
    def _param_version(self):
        return tuple(p._version for p in self.parameters()) + (id(next(self.parameters())),)

    def _resblocks(self):
        for i, m in enumerate(self.downs):
            yield f"downs.{i}.0.", m[0]
            yield f"downs.{i}.2.", m[2]
        for i, m in enumerate(self.middle):
            yield f"middle.{i}.0.", m[0]
            yield f"middle.{i}.1.", m[1]
        for i, m in enumerate(self.ups):
            yield f"ups.{i}.0.", m[0]
            yield f"ups.{i}.1.", m[1]

    @torch.no_grad()
    def prepare(self, force=False):
        """(Re)pack parameters into kernel layouts.  Called automatically when parameter versions change."""
        ver = self._param_version()
        if not force and ver == self._packed_ver:
            return
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise _lib.AidError("the MI355X network must live on the GPU (no CPU fallback); call .to('cuda')")

        def put(key, t):
            t = t.contiguous()
            if key in self._packed and self._packed[key].shape == t.shape and self._packed[key].device == t.device:
                self._packed[key].copy_(t)
            else:
                self._packed[key] = t.clone()
                self._states.clear()             # cached launch plans hold raw pointers into the replaced pack

        def ensure(key, shape):
            t = self._packed.get(key)
            if t is None or tuple(t.shape) != tuple(shape) or t.device != dev:
                t = self._packed[key] = torch.empty(*shape, device=dev, dtype=torch.float32)
                self._states.clear()
            return t

        sd = dict(self.named_parameters())
        for name, w in sd.items():
            leaf = name.rsplit(".", 1)[-1]
            if leaf == "weight" and w.dim() >= 3:                       # conv weights (2-D ones are Linears)
                wd = w.detach()
                co, ci = wd.shape[0], wd.shape[1]
                kh, kw = (wd.shape[2], wd.shape[3]) if wd.dim() == 4 else (1, wd.shape[2])
                wino = wd.dim() == 4 and (kh, kw) == (5, 3) and co >= 64 and ci >= 64       # Winograd F(4,3) packs
                if wd.dtype != torch.float32 or not wd.is_contiguous():
                    wd = wd.float().contiguous()
                (cip, cop), (cipT, copT) = _lib.pack_dims(ci, co), _lib.pack_dims(co, ci)
                w8 = wino and 8 in self.wino_forms                                            # F(8,3) packs (50 taps)
                w2 = wino and (45 in self.wino_forms or 85 in self.wino_forms) and ((co % 128 == 0 and ci % 128 == 0) or (self.w2d_c96_max_T > 0 and co % 96 == 0 and ci % 96 == 0))       # 2-D F(4,5) x F(4,3) packs (48 planes; C >= 128 layers)
                w3 = w2 and 85 in self.wino_forms                                              # ... and its F(8,3)-along-T variant (80 planes)
                bufs = [ensure(name, (kh * kw, cip, cop)), ensure(name + "#T", (kh * kw, cipT, copT)),
                        ensure(name + "#W", (30, cip, cop)) if wino else None, ensure(name + "#WT", (30, cipT, copT)) if wino else None,
                        ensure(name + "#W8", (50, cip, cop)) if w8 else None, ensure(name + "#W8T", (50, cipT, copT)) if w8 else None,
                        ensure(name + "#W2", (48, cip, cop)) if w2 else None, ensure(name + "#W2T", (48, cipT, copT)) if w2 else None,
                        ensure(name + "#W3", (80, cip, cop)) if w3 else None, ensure(name + "#W3T", (80, cipT, copT)) if w3 else None]
                pp = _lib.PackConvWeightParams(wd.data_ptr(), bufs[0].data_ptr(), bufs[1].data_ptr(), _lib.ptr(bufs[2]), _lib.ptr(bufs[3]),
                                               co, ci, kh, kw, cip, cop, cipT, copT, _lib.ptr(bufs[4]), _lib.ptr(bufs[5]), _lib.ptr(bufs[6]), _lib.ptr(bufs[7]), _lib.ptr(bufs[8]), _lib.ptr(bufs[9]))
                _lib.call("aid_pack_conv_weight", pp)       # all layouts of this weight in one launch (same values as _lib.pack_conv_weight*)
            elif leaf == "gamma":
                put(name, w.detach().reshape(-1).float())
        # ResnetBlocks that project their input twice (proj_in and res_conv on the same tensor): one stacked pack [W_pi^T ; W_rc^T / sqrt2] for the
        # merged input gradient (aid_conv2d x2 / Cin1); K segments must be multiples of 16 and start at row N of the pack
        for pfx, blk in self._resblocks():
            if hasattr(blk, "proj_in") and hasattr(blk, "res_conv") and blk.proj_place == "before" and blk.N % 16 == 0 and blk.dim >= 32 and blk.dim % 16 == 0:
                n = blk.N
                a, b = self._packed[pfx + "proj_in.weight#T"], self._packed[pfx + "res_conv.weight#T"]
                cip, cop = _lib.pack_dims(2 * n, blk.dim)
                assert a.shape[2] == cop and b.shape[2] == cop
                t = torch.zeros(1, cip, cop, device=dev, dtype=torch.float32)
                t[:, :n], t[:, n:2 * n] = a[:, :n], b[:, :n] * RSQRT2
                put(pfx + "#pairT", t)
        # stacked modulation matrix: [affine2, gate2]? then per step [affine.k, gate.k], block after block
        rows, biases, layout, off = [], [], {}, 0
        for pfx, blk in self._resblocks():
            names = []
            if blk.has_attn:
                names += ["affine2", "gate2"]
            for k in range(blk.num_dils):
                names += [f"affine.{k}", f"gate.{k}"]
            for nm in names:
                w, b = sd[pfx + nm + ".weight"], sd[pfx + nm + ".bias"]
                layout[pfx + nm] = (off, w.shape[0])
                rows.append(w.detach().float())
                biases.append(b.detach().float())
                off += w.shape[0]
        Toct = self.CQTransform.plan.T_oct
        for pfx, blk in self._resblocks():
            if not blk.has_attn:
                continue
            T = Toct[self.num_octs - blk.fdim // self.bins_per_oct]           # time length of the level this block sits on
            ab = blk.attn_block
            if hasattr(ab.qk, "bias"):                                         # bias_qkv: the conv's residual input, broadcast over T
                put(pfx + "attn_block.qk.bias#T", ab.qk.bias.detach().float()[:, None].expand(-1, T))
            if hasattr(ab, "rel_pos"):                                         # use_rel_pos: additive logit bias [H, T, T]
                put(pfx + "attn_block.rel_pos#table", ab.rel_pos.bias_table(T))
                put(pfx + "attn_block.rel_pos#bucket", ab.rel_pos.buckets(T).to(torch.int32).to(dev))
        if self.use_fencoding:
            stale = False                                                       # (the tables are copied into the input buffers at plan build:
            for i, fe in enumerate(self.freq_encodings):                        #  cached plans are dropped only when a table's VALUES changed --
                t = fe.embeddings.detach().float().reshape(2 * self.n_fenc, self.bins_per_oct)   # re-homing the parameters for the training
                old = self._packed.get(f"#fenc.{i}")                            #  step bumps every version without changing a value)
                stale = stale or old is None or old.shape != t.shape or not torch.equal(old, t.to(old.device))
                put(f"#fenc.{i}", t)
            if stale:
                self._states.clear()
        put("#modW", torch.cat(rows, 0))
        put("#modB", torch.cat(biases, 0))
        self._mod_layout, self._mod_total = layout, off
        for i in range(3):
            put(f"#emb.w{i}", self.embedding.MLP[i].weight.detach().float())
            put(f"#emb.b{i}", self.embedding.MLP[i].bias.detach().float())
        put("#emb.freq", self.embedding.RFF_freq.detach().reshape(-1).float())
        self._packed_ver = ver

    # ---------------------------------------------------------------------------------------------------
    # plan construction
    # ---------------------------------------------------------------------------------------------------
    def _mod(self, st, key):
        off, n = self._mod_layout[key]
        return st["mod"][:, off:off + n]

    def _emit_resblock(self, bd: _Builder, st, pfx: str, blk: _ResBlock, xin, yout, prev_out=None):
        B, _, F, T = xin.shape
        N, W = blk.N, self._packed
        x = xin
        pair = ((pfx + "#pairT") in W and not bd.train and self.merge_pair_dgrad
                and bool(_lib.lib().aid_conv2d_x2_supported(2 * N, N, blk.dim, F, T)))
        if hasattr(blk, "proj_in"):
            x = bd.buf(B, N, F, T)
            if pair:                                     # forward as always; the input gradients of proj_in and res_conv are taken together
                bd.conv(xin, x, W[pfx + "proj_in.weight"], blk.dim, N)
                bd.pair_dgrad(xin, x, yout, W[pfx + "#pairT"], N, blk.dim)
            else:
                bd.conv(xin, x, W[pfx + "proj_in.weight"], blk.dim, N, wpT=W[pfx + "proj_in.weight#T"], wname=pfx + "proj_in.weight")
        if blk.has_attn:
            H = blk.heads
            assert F == blk.fdim, "attention block built for a different number of frequency rows"
            sc, stb = bd.buf(B, N), bd.buf(B, 8, 2)
            bd.stats(x, W[pfx + "norm2.gamma"], self._mod(st, pfx + "affine2"), sc, stb, gname=pfx + "norm2.gamma")
            xp = bd.buf(B, H, F, T)
            bd.conv(x, xp, W[pfx + "attn_block.proj_in.weight"], N, H, in_scale=sc, wpT=W[pfx + "attn_block.proj_in.weight#T"],
                    norm_stats=stb, wname=pfx + "attn_block.proj_in.weight")
            qk = bd.buf(B, 2 * H * F, 1, T)
            qb = W.get(pfx + "attn_block.qk.bias#T")                         # bias_qkv (unet...py:321): rides on the conv's residual input
            bd.conv(xp.view(B, H * F, 1, T), qk, W[pfx + "attn_block.qk.weight"], H * F, 2 * H * F,
                    wpT=W[pfx + "attn_block.qk.weight#T"],
                    res=None if qb is None else qb.view(1, 2 * H * F, 1, T).expand(B, -1, -1, -1), res_nograd=True, wname=pfx + "attn_block.qk.weight")
            if bd.train and qb is not None:
                bd.bias_grad(qk, pfx + "attn_block.qk.bias")
            att = bd.buf(B, H, F, T)
            rpb = W.get(pfx + "attn_block.rel_pos#table")
            bd.attention(qk, xp, att, H, F, T, bias=rpb,                       # use_rel_pos (:364)
                         relpos=None if rpb is None else (W[pfx + "attn_block.rel_pos#bucket"], pfx + "attn_block.rel_pos.relative_attention_bias.weight"))
            x1 = bd.buf(B, N, F, T)
            bd.conv(att, x1, W[pfx + "attn_block.proj_out.weight"], H, N, out_scale=self._mod(st, pfx + "gate2"), res=x,
                    alpha=RSQRT2, wpT=W[pfx + "attn_block.proj_out.weight#T"], wname=pfx + "attn_block.proj_out.weight")
            x = x1
        kh, kw = blk.ks
        for k in range(blk.num_dils):
            sc, stb = bd.buf(B, N), bd.buf(B, 8, 2)
            bd.stats(x, W[pfx + f"norm.{k}.gamma"], self._mod(st, pfx + f"affine.{k}"), sc, stb, gname=pfx + f"norm.{k}.gamma")
            xn = bd.buf(B, N, F, T)
            bd.conv(x, xn, W[pfx + f"H.{k}.weight"], N, N, kh, kw, dil=(2 ** k if kh > 1 else 1), in_scale=sc, act=1,
                    out_scale=self._mod(st, pfx + f"gate.{k}"), res=x, alpha=RSQRT2, wpT=W[pfx + f"H.{k}.weight#T"],
                    norm_stats=stb, wpw=W.get(pfx + f"H.{k}.weight#W"), wpwT=W.get(pfx + f"H.{k}.weight#WT"), wname=pfx + f"H.{k}.weight",
                    wpw8=W.get(pfx + f"H.{k}.weight#W8"), wpw8T=W.get(pfx + f"H.{k}.weight#W8T"),
                    wpw2=W.get(pfx + f"H.{k}.weight#W2"), wpw2T=W.get(pfx + f"H.{k}.weight#W2T"),
                    wpw3=W.get(pfx + f"H.{k}.weight#W3"), wpw3T=W.get(pfx + f"H.{k}.weight#W3T"))
            x = xn
        if blk.proj_place == "after":
            assert hasattr(blk, "proj_out") and hasattr(blk, "res_conv")
            t1 = bd.buf(B, blk.dim_out, F, T)
            if prev_out is not None:   # (Xout + OutBlock(X))/sqrt2 folded in (unet...py:817)
                bd.conv(xin, t1, W[pfx + "res_conv.weight"], blk.dim, blk.dim_out, res=prev_out, res_scale=SQRT2,
                        wpT=W[pfx + "res_conv.weight#T"], wname=pfx + "res_conv.weight")
                a2 = 0.5
            else:
                bd.conv(xin, t1, W[pfx + "res_conv.weight"], blk.dim, blk.dim_out, wpT=W[pfx + "res_conv.weight#T"], wname=pfx + "res_conv.weight")
                a2 = RSQRT2
            bd.conv(x, yout, W[pfx + "proj_out.weight"], N, blk.dim_out, res=t1, alpha=a2, wpT=W[pfx + "proj_out.weight#T"], wname=pfx + "proj_out.weight")
        elif hasattr(blk, "res_conv") and pair:
            bd.conv(xin, yout, W[pfx + "res_conv.weight"], blk.dim, blk.dim_out, res=x, alpha=RSQRT2)
            bd.res_grad(x, yout, RSQRT2)
        elif hasattr(blk, "res_conv"):
            bd.conv(xin, yout, W[pfx + "res_conv.weight"], blk.dim, blk.dim_out, res=x, alpha=RSQRT2,
                    wpT=W[pfx + "res_conv.weight#T"], wname=pfx + "res_conv.weight")
        else:
            bd.add2(x, xin, yout, RSQRT2, RSQRT2)

    def _build_state(self, B: int, train: bool = False, lanes: bool = True, sub: bool = False):
        self.prepare()
        dev = next(self.parameters()).device
        n, bpo, Ns = self.num_octs, self.bins_per_oct, self.Ns
        assert n >= 2, "at least two octaves (the reference's pyramid path needs them too, unet...py:765-774)"
        Toct = self.CQTransform.plan.T_oct                 # low octave first
        Tl = [Toct[n - 1 - i] for i in range(n)]           # level i (0 = highest octave, longest T)
        for i in range(1, n):
            assert Tl[i] * 2 == Tl[i - 1], "octave lengths must halve per level (unet...py:768-774,786)"
        Fl = [(i + 1) * bpo for i in range(n)]
        for i in range(n):
            if self.attention_layers[i] and Tl[i] > _lib.AID_ATTN_MAX_T:
                raise NotImplementedError(f"attention at level {i} has T={Tl[i]} > {_lib.AID_ATTN_MAX_T} (aid_time_attention limit)")
        for nm in ("downsamplerT", "upsamplerT"):
            k = getattr(self, nm).kernel.detach().float().cpu()
            if not torch.allclose(k, torch.tensor(_CUBIC), atol=1e-7):
                raise NotImplementedError(f"{nm}.kernel differs from the reference's cubic FIR (unet...py:514-515), which aid_resample hard-wires")
        st = dict(B=B)
        bd = _Builder(self, B, dev, train=train)
        bd.whole_batch = bool(lanes) and not sub         # (sub-batch states are built with lanes=False -- or sub=True in the lanes_in_sub_batches A/B -- see _state)
        st["sigma"] = bd.buf(B)
        st["emb"] = bd.buf(B, self.emb_dim)
        st["mod"] = bd.buf(B, self._mod_total)
        bd.mod = st["mod"]
        if train:
            from .dist import flatten_parameters_
            flat = flatten_parameters_(self)                       # parameters as views of ONE buffer: the optimiser walks it flat
            st["gflat"] = torch.zeros_like(flat)
            names = [k for k, v in self.named_parameters() if v.dtype == torch.float32] + \
                    [k for k, v in self.named_buffers() if v.dtype == torch.float32]
            tens = [v for v in self.parameters() if v.dtype == torch.float32] + [v for v in self.buffers() if v.dtype == torch.float32]
            bd.params, bd.pgrad, off = {}, {}, 0
            for k, v in zip(names, tens):
                assert v.data_ptr() == flat.data_ptr() + 4 * off, "parameter is not where the flat buffer says it is"
                bd.params[k] = v
                bd.pgrad[k] = st["gflat"][off:off + v.numel()].view(v.shape)
                off += v.numel()
            bd.dmod = st["dmod"] = torch.zeros_like(st["mod"])
        W = self._packed
        # -- plan 0: embedding + all modulation vectors ----------------------------------------------------
        p0 = _Plan()
        ep = _lib.EmbedParams(st["sigma"].data_ptr(), W["#emb.freq"].data_ptr(), W["#emb.w0"].data_ptr(), W["#emb.b0"].data_ptr(),
                              W["#emb.w1"].data_ptr(), W["#emb.b1"].data_ptr(), W["#emb.w2"].data_ptr(), W["#emb.b2"].data_ptr(),
                              st["emb"].data_ptr(), B, W["#emb.freq"].numel(), W["#emb.w0"].shape[0], W["#emb.w1"].shape[0], self.emb_dim)
        p0.add("aid_embed", ep)
        st["embed_params"] = ep
        mp = _lib.ModulationParams(st["emb"].data_ptr(), W["#modW"].data_ptr(), W["#modB"].data_ptr(), st["mod"].data_ptr(), B,
                                   self.emb_dim, self._mod_total)
        p0.add("aid_modulation", mp)
        st["plan_mod"] = p0

        # -- buffers the CQT analysis writes (octave o feeds level n-1-o) ----------------------------------------
        pyrL = bd.buf(B, 2, n * bpo, Tl[n - 1])     # pyramid input of the deepest level = cat(C_{n-1}, pyr_{n-2}) (:773)
        if self.use_fencoding:
            # AddFreqEncodingRFF (:253-263, :754-756): octave o's init block sees [C (2 channels), 2N frozen sin/cos rows]; the
            # table is copied once into channels 2.. of a (2 + 2N)-channel input buffer, the CQT analysis writes channels 0..1
            cin_full = []
            for o in range(n):
                t = bd.buf(B, self.Nin, bpo, Toct[o])
                t[:, 2:] = W[f"#fenc.{n - 1 - o}"].view(1, 2 * self.n_fenc, bpo, 1)      # (freq_encodings[i] belongs to level i = n-1-o)
                cin_full.append(t)
            octs_in = [t[:, :2] for t in cin_full]
        else:
            cin_full = None
            octs_in = [pyrL[:, :, :bpo, :] if o == 0 else bd.buf(B, 2, bpo, Toct[o]) for o in range(n)]
        st["octs_in"] = octs_in
        D = [bd.buf(B, 2 * Ns[i], Fl[i], Tl[i]) for i in range(n)]                       # cat(X, skip) along C (:814)
        Xb = [bd.buf(B, Ns[i] if i == 0 else Ns[i - 1], Fl[i], Tl[i]) for i in range(n)]   # cat(C2, X) along F (:770)
        # -- encoder (:747-795) ---------------------------------------------------------------------------------
        pyr_prev = None
        # Lane 1 (plan.py): the per-octave init blocks, the pyramid path and the out blocks -- few-channel, latency-bound launches that depend on
        # the trunk only where they join it.  With plan.lanes = 2 they run on a second stream next to the trunk's 5x3 convolutions.
        for i in range(n):
            Cin = octs_in[n - 1 - i]
            with bd.on_lane(1):
                self._emit_resblock(bd, st, f"downs.{i}.0.", self.downs[i][0], Cin if cin_full is None else cin_full[n - 1 - i],
                                    Xb[i][:, :, :bpo, :])
                if cin_full is not None and i == n - 1:          # the deepest octave is also the first rows of the pyramid input (:773)
                    bd.copy(Cin, pyrL[:, :, :bpo, :])
                if i < n - 1:
                    pyr = pyrL[:, :, bpo:, :] if i == n - 2 else bd.buf(B, 2, Fl[i], Tl[i] // 2)
                    bd.resample(Cin, pyr[:, :, :bpo, :], up=0)
                    if i > 0:
                        bd.resample(pyr_prev, pyr[:, :, bpo:, :], up=0)
                else:
                    pyr = pyrL
            hs = D[i][:, Ns[i]:, :, :]
            self._emit_resblock(bd, st, f"downs.{i}.2.", self.downs[i][2], Xb[i], hs)
            wpyr, wpyrT = W[f"downs.{i}.1.weight"], W[f"downs.{i}.1.weight#T"]
            if i < n - 1:
                Xd = bd.buf(B, Ns[i], Fl[i], Tl[i] // 2)
                bd.resample(hs, Xd, up=0)
                bd.conv(pyr, Xb[i + 1][:, :, bpo:, :], wpyr, 2, Ns[i], 5, 3, dil=1, res=Xd, alpha=RSQRT2, wpT=wpyrT, wname=f"downs.{i}.1.weight")   # (:794)
            else:
                Xmid = bd.buf(B, Ns[i], Fl[i], Tl[i])
                bd.conv(pyr, Xmid, wpyr, 2, Ns[i], 5, 3, dil=1, res=hs, alpha=RSQRT2, wpT=wpyrT, wname=f"downs.{i}.1.weight")
            pyr_prev = pyr
        # -- bottleneck (:800-804) --------------------------------------------------------------------------------
        Xcur = Xmid
        nm = self.num_bottleneck_layers
        for m in range(nm):
            tgt = D[n - 1][:, :Ns[n - 1], :, :] if m == nm - 1 else bd.buf(B, Ns[n - 1], Fl[n - 1], Tl[n - 1])
            self._emit_resblock(bd, st, f"middle.{m}.1.", self.middle[m][1], Xcur, tgt)
            Xcur = tgt
        Xout = bd.buf(B, 2, Fl[n - 1], Tl[n - 1])
        with bd.on_lane(1):
            self._emit_resblock(bd, st, f"middle.{nm - 1}.0.", self.middle[nm - 1][0], Xcur, Xout)
        # -- decoder (:807-839) -------------------------------------------------------------------------------------
        octs_out = [None] * n
        for i in range(n):
            j = n - 1 - i
            dim_out = Ns[j - 1] if j > 0 else Ns[0]
            R = bd.buf(B, dim_out, Fl[j], Tl[j])
            self._emit_resblock(bd, st, f"ups.{i}.1.", self.ups[i][1], D[j], R)
            if j > 0:                                    # (trunk first: in the reverse sweep the out block then contributes to dL/dR before the
                bd.resample(R[:, :, bpo:, :], D[j - 1][:, :Ns[j - 1], :, :], up=1)       # trunk does, and lane 1 never waits for the trunk there)
            Xo = bd.buf(B, 2, Fl[j], Tl[j])
            with bd.on_lane(1):
                self._emit_resblock(bd, st, f"ups.{i}.0.", self.ups[i][0], R, Xo, prev_out=Xout)
                octs_out[i] = Xo[:, :, :bpo, :]
                if j > 0:
                    Xout = bd.buf(B, 2, Fl[j - 1], Tl[j - 1])
                    bd.resample(Xo[:, :, bpo:, :], Xout, up=1)
        st["octs_out"] = octs_out
        st["lanes"] = bd.plan.lanes = self._plan_lanes(B) if lanes else 1
        if st["lanes"] > 1:                              # ONE side stream for every two-lane plan of this network (they never run concurrently)
            if getattr(self, "_lane_stream", None) is None:
                self._lane_stream = torch.cuda.Stream(device=dev)
            bd.plan.side_stream = torch.cuda.Stream(device=dev) if sub else self._lane_stream      # (concurrent sub-batch plans: one side stream each)
        st["plan_body"] = bd.plan
        st["nbytes"] = bd.nbytes
        st["flops"] = bd.plan.flops
        st["keep"] = (pyrL, D, Xb, bd.stats_ws)
        st["builder"] = bd            # the input-VJP plan is emitted lazily (first guided evaluation)
        return st

    MAX_CACHED_STATES = 2      # batch sizes whose launch plans are kept alive (each owns all activations / gradients: ~61 GB at B=8 guided)

    def _state(self, B: int, slot: int = 0, group=None):
        """Launch-plan state of a (sub-)batch of B segments.  ``group`` = (total batch, number of sub-batches) the state
        belongs to (the LRU evicts whole groups); ``slot`` distinguishes equal-sized sub-batches that run concurrently."""
        self.prepare()
        group = (B, 1) if group is None else group
        grp = self._states.pop(group, None)
        if grp is None:
            while len(self._states) >= self.MAX_CACHED_STATES:          # least recently used first (dicts keep insertion order)
                self._states.pop(next(iter(self._states)))
            grp = {}
        self._states[group] = grp                                       # (re)insert as most recently used
        st = grp.get((B, slot))
        if st is None:
            st = grp[(B, slot)] = self._build_state(B, train=(group[1] == "train"), lanes=(group[1] == 1 or self.lanes_in_sub_batches), sub=(group[1] != 1 and group[1] != "train"))   # (sub-batches already share the GPU)
        return st

    # ---------------------------------------------------------------------------------------------------
    # sub-batch streams: the fused sampler entry points (denoise / denoise_guided) cut a batch into 2-3 sub-batches and
    # run them on separate HIP streams.  Segments are independent and every kernel's per-sample arithmetic is independent of the batch; the
    # Winograd form / tile instance of a 5x3 layer, however, is chosen from the LAUNCH shape (batch included), so a split agrees with the unsplit
    # schedule to the bit only where both take the same instances (small networks) and to rounding (<= 5e-6, tests/test_gpu_dist.py) at full size.
    # What changes is the schedule: while one sub-batch is in an MFMA-bound conv, the HBM-bound passes (statistics, activation / transform
    # pre-passes, normalisation backward) and the partially filled last round of workgroups of another sub-batch's conv use
    # the idle wave slots and CUs (measured at B=8, guided: +6.3 % with 2 sub-batches, +7.6 % with 3, -1.7 % with 4).
    # ---------------------------------------------------------------------------------------------------
    split_streams = None       # None: automatic (2 sub-batches for B >= 4, else 1); an int forces it.  (Round 4: with the 512-position tiles of the F(8,3)
                               # kernel two sub-batches of four beat three of 3 / 3 / 2 -- 46.8 vs 44.9 evaluations/s at batch 8, profiles/r04_bench_after_form_model.txt;
                               # with F(4,3) alone it was 43.5 either way.)

    cu_partition = 0           # EXPERIMENT (round 6; bench.py --cu-partition N): > 0 = spatial partition of the chip for the sub-batch
                               # streams -- inside every sub-batch the launch plans run on TWO CU-masked streams (streams.py), lane 0 = the MFMA-bound kernels
                               # (5x3 / 1x1 convs above the fp32 ridge, the 2-D form's GEMM, attention) on 256 - N CUs, lane 1 = the HBM-bound passes on the
                               # last N / 8 CUs of every XCD; the cross-lane dependencies are the plan's derived ones (plan.py).  0 = free-running streams.
    cu_partition_ridge = 30.0  # FLOP per algorithmic byte above which a conv launch counts as MFMA-bound (fp32 ridge ~ 20-26 FLOP/B)

    cu_split = None            # EXPERIMENT (bench.py --cu-split a,b[,c]): every sub-batch stream gets its OWN share of each XCD's 32 CUs (counts per XCD, e.g. 16,16 or
                               # 20,12 with --split 5,3; shares may overlap when they sum to more than 32: the first starts at CU 0, the last ends at CU 31) -- no
                               # cross-stream events at all: each sub-batch's chain runs undisturbed on its partition.  None = unmasked free-running streams.

    def _split_mask_streams(self, n: int):
        key = (tuple(self.cu_split), n)
        if getattr(self, "_csplit_key", None) != key:
            from .streams import cu_masked_stream, xcd_range_mask
            shares = [int(v) for v in self.cu_split]
            assert len(shares) == n and all(0 < v <= 32 for v in shares), "cu_split: one CU count per XCD (1..32) for every sub-batch stream"
            total = sum(shares)
            starts, pos = [], 0
            for k, v in enumerate(shares):                 # spread the shares over [0, 32): disjoint when they fit, evenly overlapping when they do not
                starts.append(0 if n == 1 else round(k * (32 - v) / (n - 1)) if total > 32 else pos)
                pos += v
            self._csplit_streams = [cu_masked_stream(xcd_range_mask(st, st + v)) for st, v in zip(starts, shares)]
            self._csplit_key = key
        return self._csplit_streams

    def _type_lane(self, op) -> int:
        if op.name in ("aid_conv2d_wino2d_gemm", "aid_time_attention", "aid_time_attention_bwd"):
            return 0
        if op.name == "aid_conv2d" and op.flops > 0 and op.nbytes > 0 and op.flops / op.nbytes >= self.cu_partition_ridge:
            return 0
        return 1

    def _partition_streams(self, n: int):
        """[(MFMA stream, pass stream)] per sub-batch, CU-masked (created once per partition size)"""
        key = (int(self.cu_partition), n)
        if getattr(self, "_part_key", None) != key:
            from .streams import cu_masked_stream, partition_masks
            mw, pw = partition_masks(int(self.cu_partition))
            self._part_streams = [(cu_masked_stream(mw), cu_masked_stream(pw)) for _ in range(n)]
            self._part_key = key
        return self._part_streams

    def _retag_by_type(self, plan, side):
        """lanes of a launch plan by kernel type (see cu_partition); dependencies are re-derived by plan.schedule()"""
        if getattr(plan, "_typed", False) is False:
            for op in plan.ops:
                op.lane = self._type_lane(op)
            plan._sched = None
            plan._typed = True
        plan.lanes, plan.side_stream = 2, side

    def _n_split(self, B: int) -> int:
        n = self.split_streams
        if n is None:
            n = 2 if B >= 4 else 1
        return max(1, min(int(n), B))

    split_sizes = None         # experiments: explicit sub-batch sizes, e.g. (5, 3) for a batch of 8 (None: equal shares)

    def _split_plan(self, B: int):
        n = self._n_split(B)
        bounds = [(i * B) // n for i in range(n + 1)]
        if self.split_sizes is not None and sum(self.split_sizes) == B and len(self.split_sizes) == n:
            bounds = [0]
            for k in self.split_sizes:
                bounds.append(bounds[-1] + int(k))
        if self.cu_partition:
            return n, bounds, [m for m, _ in self._partition_streams(n)]
        if self.cu_split:
            return n, bounds, self._split_mask_streams(n)
        if getattr(self, "_side_streams", None) is None or len(self._side_streams) < n:
            # (equal priorities: raising one or two of the three sub-batch streams measured 43.7 -> 41.6 ... 42.7 evaluations/s)
            self._side_streams = [torch.cuda.Stream() for _ in range(n)]
        return n, bounds, self._side_streams[:n]

    def states_of(self, B: int):
        """All launch-plan states the fused entry points use for a batch of B (one per sub-batch)."""
        n, bounds, _ = self._split_plan(B) if self._n_split(B) > 1 else (1, [0, B], None)
        return [self._state(bounds[i + 1] - bounds[i], i, (B, n)) for i in range(n)]

    # ---------------------------------------------------------------------------------------------------
    # execution
    # ---------------------------------------------------------------------------------------------------
    def _check_input(self, inputs):
        if not inputs.is_cuda:
            raise _lib.AidError("Unet_CQT_oct_with_attention (MI355X build) needs CUDA/HIP tensors; there is no CPU fallback")
        if inputs.dim() != 2 or inputs.shape[-1] != self.CQTransform.Ls:
            raise ValueError(f"expected inputs of shape [B, {self.CQTransform.Ls}], got {tuple(inputs.shape)}")

    def _run_body(self, st, sigma):
        B = st["B"]
        st["sigma"].copy_(sigma.reshape(-1).to(torch.float32).expand(B) if sigma.numel() == 1 else sigma.reshape(B).to(torch.float32))
        st["plan_mod"].run()
        if self.cu_partition and st.get("part_slot") is not None:
            self._retag_by_type(st["plan_body"], self._part_streams[st["part_slot"]][1])
        st["plan_body"].run()

    def forward(self, inputs: torch.Tensor, sigma: torch.Tensor) -> torch.Tensor:
        """inputs[B,L] (time domain), sigma[B,1] (= c_noise) -> [B,L]     (unet...py:730-845)"""
        self._check_input(inputs)
        if torch.is_grad_enabled() and inputs.requires_grad and not (self.training and self.param_grads_in_train_mode):
            from .autograd import DenoiserFn   # input-VJP through the same kernels (guidance branch; the reference's tester leaves
            if self.training and not getattr(self, "_warned_input_only", False) and any(p.requires_grad for p in self.parameters()):
                import warnings                # the network in train() mode, so this case comes first -- and says so once)
                self._warned_input_only = True
                warnings.warn("Unet_CQT_oct_with_attention: train()-mode call with an input that requires grad -> input-VJP only (what the "
                              "sampler's guidance needs); parameter gradients are NOT produced on this path. Set "
                              "net.param_grads_in_train_mode = True to get both.", RuntimeWarning, stacklevel=2)
            return DenoiserFn.apply(inputs, sigma, self)
        if torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters()):
            from .autograd import TrainFn      # training: gradients w.r.t. the parameters (and the input, if it asks) through the same kernels
            return TrainFn.apply(inputs, sigma, self, *self.parameters())
        return self._forward_impl(inputs, sigma)

    @torch.no_grad()
    def _forward_impl(self, inputs, sigma):
        B, L = inputs.shape
        st = self._state(B)
        x = inputs.detach().contiguous().float()
        self.CQTransform.analysis(x, st["octs_in"])
        self._run_body(st, sigma)
        Y = self.CQTransform.synthesis_spectrum(st["octs_out"])
        return self.CQTransform.irfft(Y)

    # ---------------------------------------------------------------------------------------------------
    # HIP-graph replay of launch-bound evaluations.  A fused evaluation is ~1000 (forward) / ~1900 (guided) kernel
    # launches from pre-filled parameter structs; at B <= 3 the GPU finishes them faster than the host can issue them
    # (the reference's testers run B = 1).  The first call of a configuration runs eagerly (lazy plan construction, table
    # uploads, hipFuncSetAttribute), the second captures the same call sequence on torch's capture stream with static
    # input / output buffers, later calls copy the inputs in and replay.  Sub-batch streams (B >= 4) stay eager.
    # ---------------------------------------------------------------------------------------------------
    use_graphs = False         # OFF by default since round 5: with three launches per 2-D Winograd layer, most of them long, eager launches on the two lanes beat the
                               # replay at every small batch -- batch 1 / 2 / 3: 36.9 / 44.9 / 47.9 with replay, 37.5 / 45.5 / 48.5 without (two alternating runs each,
                               # profiles/r05_graphs_ab.txt; round 3, when an evaluation was ~1900 short launches: +8 %).  `net.use_graphs = True` (bench.py --graphs)
                               # turns it back on, e.g. for hosts that cannot issue ~30 000 launches per second per rank.
    fuse_dot_1x1 = True        # reverse sweep: <gd, x> partials of the 1x1 steps from the dgrad conv's epilogue instead of an aid_group_dot pass
    merge_pair_dgrad = True    # reverse sweep: input gradients of a block's proj_in and res_conv as ONE 1x1 conv over a K axis in two tensors
    plan_lanes = True          # tag the init blocks / pyramid / out blocks as lane 1 of the launch plans (plan.py)
    lanes_in_sub_batches = False   # A/B: two-lane plans inside sub-batch streams too (bench.py --lanes-in-sub-batches --lanes-max-batch 4): 56.4 -> 55.8 at batch 8
                                   # (profiles/r05_lanes_ab.txt); one unsplit batch of 8 on two lanes: 54.4
    lanes_max_batch = 3        # ... and run the two lanes on two streams for whole batches up to this size.  Larger batches fill the GPU and run as
                               # sub-batch streams; a second lane INSIDE each sub-batch stream measured -19 % at batch 8 (six streams competing)
    param_grads_in_train_mode = False   # True: a train()-mode call whose INPUT requires grad also yields parameter gradients (TrainFn) instead of
                                        # the input-only VJP; off by default because the reference's tester samples with the network in train() mode
    wino_forms = (4, 8, 45, 85)    # 85 (round 6) = the 2-D form with F(8,3) along T, F(4,5) x F(8,3): 80 products per 4 x 8 outputs (2.5 per output), where the library
                               # prefers it over 45 (aid_conv2d_wino2d_tform).  Winograd forms the 5x3 layers may use: 45 = the non-fused 2-D form F(4,5) x F(4,3) (csrc/aid_wino2d.hip) on the C >= 128 layers
                               # where the library predicts it faster (aid_conv2d_wino2d_wanted), F(8,3) where the library prefers it (aid_conv2d_wino_form),
                               # F(4,3) otherwise; (4,) keeps every layer on the F(4,3) kernels, (45,) forces the 2-D form wherever it is supported
                               # (A/B measurements, tests; set before the first forward)
    form_batch = None          # None: the Winograd FORM of a 5x3 layer (2-D / F(8,3) / F(4,3)) is chosen for the batch of the launch it runs in -- the fastest, and the reason
                               # why a sub-batch split or another world size reproduces a segment to rounding (<= 5e-6 per evaluation) and not to the bit.
                               # An int pins the choice to that batch for every launch (e.g. 1: the forms of a single-segment launch everywhere), so the FORM no longer
                               # depends on how segments are grouped; the tile INSTANCE inside a form (K-group / split-K instances for launches with few tiles) still
                               # follows the launch shape, so this narrows the difference, it does not promise identical bits.  Set before the first forward.
    w2d_c96_max_T = 1024       # the 96-channel levels with T up to this take the 2-D form where the library wants it (96 x 128 GEMM tiles, 80 planes; round 6);
                               # 0: never (A/B, bench.py --w2d-c96-max-t 0; set before the first forward: it also decides which packs are built)
    w2d_force_max_T = 0        # A/B: layers with T up to this take the 2-D form wherever it is SUPPORTED, whatever the library's per-layer prediction says (bench.py --w2d-force-max-t)
    w2d_min_channels = 96      # A/B: 256 keeps the 2-D form off the K = 128 levels the library would give it (bench.py --w2d-min-channels)
    wgrad_wino = True          # training: F(4,3) form of the 5x3 weight gradients (aid_conv2d_wgrad wino=1)
    fuse_fin = True            # the last tile of a sample folds the conv epilogue's statistics / dot partials itself (aid_conv2d fin_mode): no aid_group_stats
                               # launch after such a conv and no coefficient kernel in aid_norm_bwd (A/B: bench.py --no-fin)
    fold_grad_copies = True    # reverse sweep: a first-contribution scaled copy dL/dres = c dL/dy is not launched where the last dilated step of a block can read
                               # dL/dy through it (aid_scale_act mul / aid_norm_bwd a): _Builder._defer_copy (A/B: bench.py --no-fold-copies)
    fuse_norm_bwd_wino = True  # reverse sweep: aid_norm_bwd also writes the Winograd-domain, gated copy that the dgrad conv below stages
    epilogue_stats = True      # forward group statistics from the epilogue of the conv that produces the tensor (row-shared F(4,3) kernel)
    GRAPH_MAX_B = 3
    GRAPH_MAX_PER_STATE = 4

    def _plan_lanes(self, B: int) -> int:
        return 2 if (self.plan_lanes and B <= self.lanes_max_batch) else 1

    def _graph_ok(self, B, st):
        if not (self.use_graphs and B <= self.GRAPH_MAX_B and self._n_split(B) == 1):
            return False
        plans = [st["plan_body"]] + ([st["plan_bwd"]] if "plan_bwd" in st else [])
        return all(pl.timing is None for pl in plans)

    def _graph_run(self, st, key, inputs, scalars, fn):
        """inputs: dict name -> tensor copied into static buffers; scalars: (cnoise, cin, cskip, cout) as host floats or device [B]
        tensors, written into a static [4, B] buffer outside the graph; fn(static dict) -> tuple of output tensors.
        Returns clones of the static outputs."""
        graphs = st.setdefault("graphs", {})
        ent = graphs.get(key)
        B = st["B"]
        dev = next(iter(inputs.values())).device
        names = ("cnoise", "cin", "cskip", "cout")
        if ent is None:                                   # first call: eager (builds lazily created plans / tables)
            graphs[key] = {"seen": 1}
            while len(graphs) > self.GRAPH_MAX_PER_STATE:
                graphs.pop(next(iter(graphs)))
            d = dict(inputs)
            d.update(zip(names, self._scalar_rows(B, dev, *scalars)))
            return fn(d)

        def put_scalars(buf):
            if all(isinstance(v, float) for v in scalars):
                self._scalar_rows(B, dev, *scalars, out=buf)
            else:
                for i, v in enumerate(scalars):
                    buf[i].copy_(v.reshape(-1))
        if "graph" not in ent:                            # second call: capture
            static = {k: v.clone() for k, v in inputs.items()}
            sbuf = torch.empty(4, B, device=dev, dtype=torch.float32)
            put_scalars(sbuf)
            static.update(zip(names, (sbuf[0], sbuf[1], sbuf[2], sbuf[3])))
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                outs = fn(static)
            ent.update(graph=g, static=static, outs=outs, sbuf=sbuf)
        else:
            for k, v in inputs.items():
                ent["static"][k].copy_(v)
            put_scalars(ent["sbuf"])
        ent["graph"].replay()
        return tuple(o.clone() for o in ent["outs"])

    @staticmethod
    def _norm_scalars(B, device, vals):
        """Validate the four EDM scalars of a public entry point: every one a real number / 0-d or 1-element value (-> host float, one noise
        level for the batch) or every one a float32 tensor of B elements on ``device`` (a 1-element tensor is expanded).  Anything else --
        host pointers, wrong lengths, mixed forms that cannot be reconciled -- raises AidError instead of reaching a kernel."""
        import numbers
        out, kinds = [], set()
        for v in vals:
            if isinstance(v, numbers.Real) or (hasattr(v, "ndim") and not torch.is_tensor(v) and getattr(v, "size", 2) == 1):
                out.append(float(v)); kinds.add("h")
            elif torch.is_tensor(v) and v.numel() == 1 and v.device.type != "cuda":
                out.append(float(v)); kinds.add("h")
            elif torch.is_tensor(v):
                if v.numel() not in (1, B):
                    raise _lib.AidError(f"EDM scalar with {v.numel()} elements for a batch of {B}")
                t = v.detach().reshape(-1).to(device=device, dtype=torch.float32)
                out.append((t.expand(B) if t.numel() == 1 else t).contiguous()); kinds.add("d")
            else:
                raise _lib.AidError(f"EDM scalar of unsupported type {type(v).__name__}")
        if kinds == {"h", "d"}:                             # mixed: lift the host values to device vectors
            out = [torch.full((B,), v, dtype=torch.float32, device=device) if isinstance(v, float) else v for v in out]
        return tuple(out)

    def _scalar_rows(self, B, device, cnoise, cin, cskip, cout, out=None):
        """The four per-evaluation EDM scalars as device [B] vectors.  Host floats (one noise level for the whole batch -- what the sampling
        loop has) are broadcast by ONE aid_set_rows launch into a persistent [4, B] buffer; device tensors pass through."""
        vals = (cnoise, cin, cskip, cout)
        if all(isinstance(v, float) for v in vals):
            buf = out
            if buf is None:
                cache = self.__dict__.setdefault("_scal_cache", {})
                buf = cache.get((B, device))
                if buf is None:
                    buf = cache[(B, device)] = torch.empty(4, B, device=device, dtype=torch.float32)
            sp = _lib.SetRowsParams(buf.data_ptr(), buf.stride(0), B, 4, (C.c_float * 8)(*vals, 0.0, 0.0, 0.0, 0.0))
            _lib.call("aid_set_rows", sp)
            return buf[0], buf[1], buf[2], buf[3]
        return tuple(v.reshape(-1) for v in vals)

    @torch.no_grad()
    def denoise(self, x, cnoise, cin, cskip, cout, hpf: bool):
        """Fused EDM denoiser  D(x) = [hpf](cskip*x + cout*F(cin*x, cnoise))  (diff_params/edm.py:133-148 and,
        with hpf=True, CQT.apply_hpf_DC of edm_sampler_inpainting.py:63).  cin/cskip/cout/cnoise: device [B] tensors, or four host
        floats (one noise level for the batch)."""
        self._check_input(x)
        B, L = x.shape
        x = x.contiguous()
        cnoise, cin, cskip, cout = self._norm_scalars(B, x.device, (cnoise, cin, cskip, cout))
        if self._n_split(B) == 1:
            st = self._state(B)
            if self._graph_ok(B, st):
                return self._graph_run(st, ("fwd", bool(hpf)), dict(x=x), (cnoise, cin, cskip, cout),
                                       lambda t: (self._denoise_one(t["x"], t["cnoise"], t["cin"], t["cskip"], t["cout"], hpf, st),))[0]
            cnoise, cin, cskip, cout = self._scalar_rows(B, x.device, cnoise, cin, cskip, cout)
            return self._denoise_one(x, cnoise, cin, cskip, cout, hpf, st)
        cnoise, cin, cskip, cout = self._scalar_rows(B, x.device, cnoise, cin, cskip, cout)
        n, bounds, streams = self._split_plan(B)
        out = torch.empty(B, L, device=x.device, dtype=torch.float32)
        cur = torch.cuda.current_stream()
        for i in range(n):
            lo, hi = bounds[i], bounds[i + 1]
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                sti = self._state(hi - lo, i, (B, n))
                sti["part_slot"] = i if self.cu_partition else None
                self._denoise_one(x[lo:hi], cnoise[lo:hi], cin[lo:hi], cskip[lo:hi], cout[lo:hi], hpf, sti, out=out[lo:hi])
        for st_ in streams:
            cur.wait_stream(st_)
        return out

    def _denoise_one(self, x, cnoise, cin, cskip, cout, hpf, st, out=None):
        X = self.CQTransform.analysis(x, st["octs_in"], in_scale=cin)
        self._run_body(st, cnoise)
        Y = self.CQTransform.synthesis_spectrum(st["octs_out"], X=X, cskip=cskip, cout=cout, hpf=hpf)
        return self.CQTransform.irfft(Y, out=out)

    # ---------------------------------------------------------------------------------------------------
    # input-VJP (reconstruction guidance, testing/edm_sampler_inpainting.py:57-105)
    # ---------------------------------------------------------------------------------------------------
    def _bwd_plan(self, st):
        if "plan_bwd" not in st:
            bd = st["builder"]
            st["plan_bwd"] = bd.finish_backward()
            st["plan_bwd"].lanes = st["lanes"]
            st["plan_bwd"].side_stream = st["plan_body"].side_stream
            st["gin"] = [bd.G(t) for t in st["octs_in"]]
            st["gout"] = [bd.G(t) for t in st["octs_out"]]
            st["gzero"] = [g for k, g in bd.gmap.items() if bd.gstate.get(k) != "full"]
            st["nbytes"] = bd.nbytes
        return st["plan_bwd"]

    @torch.no_grad()
    def _body_vjp(self, st, gYsum):
        """gYsum[B,Lh] complex: gradient w.r.t. the synthesis band-sum spectrum of the LAST forward of this batch
        size -> fills the gradients of the analysis octave tensors (st['gin'])."""
        plan = self._bwd_plan(st)
        if self.cu_partition and st.get("part_slot") is not None:
            self._retag_by_type(plan, self._part_streams[st["part_slot"]][1])
        for g in st["gzero"]:
            g.zero_()
        self.CQTransform.synthesis_adjoint(gYsum, st["gout"])
        plan.run()

    @torch.no_grad()
    def vjp(self, g_pred: torch.Tensor) -> torch.Tensor:
        """d<g_pred, forward(inputs, sigma)>/d inputs for the most recent ``forward`` of this batch size."""
        B, L = g_pred.shape
        st = self._state(B)
        tr = self.CQTransform
        tab = tr._tables(g_pred.device)
        G = tr.rfft(g_pred.detach().float().contiguous())
        self._body_vjp(st, tr.spectrum_scale(G, tab["w_over_L"]))
        S = tr.analysis_adjoint(st["gin"])
        return tr.irfft(S)

    @torch.no_grad()
    def denoise_guided(self, x, cnoise, cin, cskip, cout, hpf: bool, y, mask, degradation=None, norm_type=2, beta=1.0):
        """Fused guided evaluation: x_hat = [hpf](cskip*x + cout*F(cin*x)) and
        rec_grads = d/dx || y - A(x_hat) ||_2 (per item, edm_sampler_inpainting.py:60-81), computed with the
        hand-written input-VJP instead of torch.autograd.  A = time-domain mask (default) or any linear operator
        object with ``apply`` / ``adjoint`` (stft.SpectralMask).  norm_type: tester.posterior_sampling.norm -- 2, 1 or "smoothl1"
        (beta = smoothl1_beta), all seeded analytically by aid_guidance_seed (:72-75).  cnoise .. cout: device [B] tensors or host floats.
        Returns (x_hat, rec_grads, norm[B])."""
        nt = {2: 2, 1: 1, "smoothl1": 3, 3: 3}.get(norm_type)
        if nt is None:
            raise _lib.AidError(f"denoise_guided: posterior_sampling.norm={norm_type!r} is not one of 2, 1, 'smoothl1'")
        nk = (nt, float(beta))
        self._check_input(x)
        B, L = x.shape
        x = x.contiguous()
        cnoise, cin, cskip, cout = self._norm_scalars(B, x.device, (cnoise, cin, cskip, cout))
        if not (y.is_cuda and y.dtype == torch.float32 and y.is_contiguous() and y.shape[0] == B and (degradation is not None or tuple(y.shape) == (B, L))):
            raise _lib.AidError("denoise_guided: y must be a contiguous float32 GPU tensor of shape [B, L] ([B, ...] with a degradation operator)")
        if (degradation is not None and not getattr(degradation, "shared_mask", False)) or self._n_split(B) == 1:
            st = self._state(B)                                      # (per-item operator masks: one stream)
            if degradation is None and self._graph_ok(B, st):
                key = ("guided", bool(hpf), y.data_ptr(), mask.data_ptr(), tuple(mask.shape), nk)   # y / mask are read in place
                return self._graph_run(st, key, dict(x=x), (cnoise, cin, cskip, cout), lambda t: self._denoise_guided_one(
                    t["x"], t["cnoise"], t["cin"], t["cskip"], t["cout"], hpf, y, mask, None, st, nk=nk))
            cnoise, cin, cskip, cout = self._scalar_rows(B, x.device, cnoise, cin, cskip, cout)
            return self._denoise_guided_one(x, cnoise, cin, cskip, cout, hpf, y, mask, degradation, st, nk=nk)
        cnoise, cin, cskip, cout = self._scalar_rows(B, x.device, cnoise, cin, cskip, cout)
        m = None if degradation is not None else (mask if mask.dim() == 2 else mask.reshape(1, -1))
        n, bounds, streams = self._split_plan(B)
        x_hat = torch.empty(B, L, device=x.device, dtype=torch.float32)
        grads = torch.empty(B, L, device=x.device, dtype=torch.float32)
        norm = torch.empty(B, device=x.device, dtype=torch.float32)
        cur = torch.cuda.current_stream()
        for i in range(n):
            lo, hi = bounds[i], bounds[i + 1]
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                sti = self._state(hi - lo, i, (B, n))
                sti["part_slot"] = i if self.cu_partition else None
                self._denoise_guided_one(x[lo:hi], cnoise[lo:hi], cin[lo:hi], cskip[lo:hi], cout[lo:hi], hpf, y[lo:hi],
                                         None if m is None else (m[lo:hi] if m.shape[0] > 1 else m), degradation, sti,
                                         outs=(x_hat[lo:hi], grads[lo:hi], norm[lo:hi]), nk=nk)
        for st_ in streams:
            cur.wait_stream(st_)
        return x_hat, grads, norm

    def _denoise_guided_one(self, x, cnoise, cin, cskip, cout, hpf, y, mask, degradation, st, outs=None, nk=(2, 1.0)):
        B, L = x.shape
        tr = self.CQTransform
        tab = tr._tables(x.device)
        X = tr.analysis(x, st["octs_in"], in_scale=cin)
        self._run_body(st, cnoise)
        Y = tr.synthesis_spectrum(st["octs_out"], X=X, cskip=cskip, cout=cout, hpf=hpf)
        x_hat = tr.irfft(Y, out=None if outs is None else outs[0])
        g = torch.empty_like(x_hat)
        norm = torch.empty(B, device=x.device, dtype=torch.float32) if outs is None else outs[2]
        if degradation is None:
            m = mask if mask.dim() == 2 else mask.reshape(1, -1)
            if not (m.is_cuda and m.dtype == torch.float32 and m.stride(-1) == 1 and m.shape[-1] == L and m.shape[0] in (1, B)):
                raise _lib.AidError("denoise_guided: mask must be a float32 GPU tensor [1|B, L] with unit inner stride "
                                    "(Sampler.setup_inpainting normalises it)")
            sp = _lib.GuidanceSeedParams(x_hat.data_ptr(), y.data_ptr(), m.data_ptr(), m.stride(0) if m.shape[0] > 1 else 0,
                                         g.data_ptr(), norm.data_ptr(), B, L, nk[0], nk[1])
            _lib.call("aid_guidance_seed", sp)
        else:                                    # g = -A^T d norm(y - A x_hat) / d r   (no mask: the operator is the degradation)
            den = degradation.apply(x_hat)
            Ld = den.numel() // B                # observations may live in another space than x_hat (Sampler.predict_resample: any per-item shape)
            if not (den.is_cuda and den.dtype == torch.float32 and den.is_contiguous() and den.shape[0] == B and y.numel() == den.numel()):
                raise _lib.AidError(f"denoise_guided: degradation(x_hat) {tuple(den.shape)} must be a contiguous float32 GPU tensor of the "
                                    f"observations' shape {tuple(y.shape)}")
            gd = g if Ld == L else torch.empty_like(den)
            sp = _lib.GuidanceSeedParams(den.data_ptr(), y.data_ptr(), None, 0, gd.data_ptr(), norm.data_ptr(), B, Ld, nk[0], nk[1])
            _lib.call("aid_guidance_seed", sp)
            g = degradation.adjoint(gd.view(den.shape))
            if not (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and tuple(g.shape) == (B, L)):
                raise _lib.AidError("denoise_guided: degradation.adjoint must return a contiguous float32 GPU tensor [B, L]")
        Gh = tr.rfft(g)
        if hpf:
            Gh = tr.spectrum_scale(Gh, tab["hpf"])                       # the projector is self-adjoint
        self._body_vjp(st, tr.spectrum_scale(Gh, tab["w_over_L"], per_item=cout))
        S = tr.analysis_adjoint(st["gin"], in_scale=cin, X=Gh, cskip=cskip)
        return x_hat, tr.irfft(S, out=None if outs is None else outs[1]), norm

    # ---------------------------------------------------------------------------------------------------
    # training step support: loss + parameter gradients on the same kernels
    # ---------------------------------------------------------------------------------------------------
    def train_state(self, B: int):
        return self._state(B, 0, (B, "train"))

    def _tail_plan(self, st):
        """modulation / embedding backward + hand-out of the stacked modulation gradients to the individual Linears."""
        if "plan_tail" in st:
            return st["plan_tail"]
        bd, W, E, B = st["builder"], self._packed, self.emb_dim, st["B"]
        N = self._mod_total
        st["dWm"], st["dbm"] = bd.buf(N, E), bd.buf(N)
        st["demb"] = bd.buf(B, E)
        pl = _Plan()
        st["mpart"] = bd.buf(B * (-(-N // 128)) * E)
        mp = _lib.ModulationBwdParams(st["dmod"].data_ptr(), st["emb"].data_ptr(), W["#modW"].data_ptr(), st["dWm"].data_ptr(),
                                      st["dbm"].data_ptr(), st["demb"].data_ptr(), B, E, N, 0, st["mpart"].data_ptr(), st["mpart"].numel())
        pl.add("aid_modulation_bwd", mp)
        g = bd.pgrad
        ep = _lib.EmbedBwdParams(st["embed_params"], st["demb"].data_ptr(),
                                 g["embedding.MLP.0.weight"].data_ptr(), g["embedding.MLP.0.bias"].data_ptr(),
                                 g["embedding.MLP.1.weight"].data_ptr(), g["embedding.MLP.1.bias"].data_ptr(),
                                 g["embedding.MLP.2.weight"].data_ptr(), g["embedding.MLP.2.bias"].data_ptr(), 0)
        pl.add("aid_embed_bwd", ep)
        st["embed_bwd_params"] = ep
        st["plan_tail"] = pl
        return pl

    @torch.no_grad()
    def _train_forward(self, inputs, sigma):
        """forward on the training state (its backward plan also carries the parameter-gradient ops)"""
        self._check_input(inputs)
        st = self.train_state(inputs.shape[0])
        x = inputs.detach().contiguous().float()
        self.CQTransform.analysis(x, st["octs_in"])
        self._run_body(st, sigma)
        return self.CQTransform.irfft(self.CQTransform.synthesis_spectrum(st["octs_out"]))

    @torch.no_grad()
    def _train_backward(self, g_out, need_input=False, accumulate=False):
        """g_out[B,L] = d loss / d network output of the LAST _train_forward -> fills the flat gradient buffer (accumulate: adds to what an
        earlier accumulation round left there, trainer.py:259-266); returns (d loss / d inputs or None, [gradient view per parameter, in
        self.parameters() order])."""
        B = g_out.shape[0]
        st = self.train_state(B)
        tr = self.CQTransform
        tab = tr._tables(g_out.device)
        Gh = tr.rfft(g_out.detach().float().contiguous())
        if not accumulate:
            st["gflat"].zero_()
        st["dmod"].zero_()
        self._body_vjp(st, tr.spectrum_scale(Gh, tab["w_over_L"]))
        tail = self._tail_plan(st)
        st["embed_bwd_params"].accumulate = 1 if accumulate else 0
        tail.run()
        gp = st["builder"].pgrad
        for key, (off, n) in self._mod_layout.items():               # stacked [sum N, E] rows -> the Linears' own gradient tensors
            if accumulate:
                gp[key + ".weight"].add_(st["dWm"][off:off + n])
                gp[key + ".bias"].add_(st["dbm"][off:off + n])
            else:
                gp[key + ".weight"].copy_(st["dWm"][off:off + n])
                gp[key + ".bias"].copy_(st["dbm"][off:off + n])
        gin = tr.irfft(tr.analysis_adjoint(st["gin"])) if need_input else None
        return gin, [gp[k] if p.requires_grad else None for k, p in self.named_parameters()]

    @torch.no_grad()
    def loss_and_grads(self, inputs: torch.Tensor, cnoise: torch.Tensor, target: torch.Tensor, hpf_error: bool = False, fir=None,
                       accumulate: bool = False):
        """error = net(inputs, cnoise) - target  [-> apply_hpf_DC(error) if hpf_error, edm.py:180-187] [-> fir.apply(error): the A-weighting
        filter, edm.py:189-190];  loss = mean(error**2) (trainer.py:262-263).  Fills (accumulate: adds to) the flat gradient buffer of the
        training state (``train_state(B)['gflat']``, laid out like the flat parameter buffer) with d loss / d parameter, all through HIP
        kernels.  Returns (loss [device scalar], error**2)."""
        B, L = inputs.shape
        tr = self.CQTransform
        est = self._train_forward(inputs, cnoise)
        err = torch.empty_like(est)
        minus1 = torch.full((B,), -1.0, device=est.device)
        _lib.call("aid_axpby", _lib.AxpbyParams(est.data_ptr(), target.contiguous().float().data_ptr(), err.data_ptr(), None, minus1.data_ptr(), B, L))
        if hpf_error:
            err = tr._hpf(err)
        if fir is not None:
            err = fir.apply(err)
        rn = torch.empty(B, device=est.device, dtype=torch.float32)
        _lib.call("aid_row_norm", _lib.RowNormParams(err.data_ptr(), rn.data_ptr(), B, L))
        loss = (rn * rn).sum() / (B * L)
        # seed: d loss / d estimate = 2 error / (B L)   (through the self-adjoint DC/Nyquist projector once more if it was applied)
        g = torch.empty_like(err)
        coef = torch.full((B,), 2.0 / (B * L), device=est.device)
        _lib.call("aid_axpby", _lib.AxpbyParams(err.data_ptr(), None, g.data_ptr(), coef.data_ptr(), None, B, L))
        if fir is not None:
            g = fir.adjoint(g)
        if hpf_error:
            g = tr._hpf(g)
        self._train_backward(g, accumulate=accumulate)
        return loss, err * err

    def flops_per_eval(self, B: int = 1) -> int:
        """Algorithmic conv/GEMM/attention FLOPs of one forward evaluation at batch B (2*MACs)."""
        return self._state(B)["flops"]

    def timed_plans(self, B: int, guided: bool):
        """The launch plans a fused evaluation of batch B runs (forward and, if guided, input-VJP, of every sub-batch) -- for
        bench.py, which sets their ``timing`` lists."""
        plans = []
        for st in self.states_of(B):
            plans.append(st["plan_body"])
            if guided:
                plans.append(self._bwd_plan(st))
        return plans
