"""BasicEncoder.forward on the HIP path: the host side of include/ctk.h's encoder entry points.

Strings ctk_conv2d_sh / ctk_enc_* together in the order of the reference (cotracker/models/core/cotracker/blocks.py:
BasicEncoder.forward :184-219, ResidualBlock.forward :128-138) for the ``fnet`` of a model, reading its parameters
(``fnet.*`` state_dict keys are unchanged: the weights are repacked per device, like the Linear layers).  No MIOpen, no
torch convolution: torch only owns the memory.  Output = what ``model._encode`` produces with the torch encoder: the
L2-normalised NHWC level-0 features [T, H/4, W/4, 128] (``normalize=False``: the raw ``conv3`` output, what CoTracker2 tracks on).
"""
import ctypes as C

import torch

from . import _lib as L
from . import ops


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class _Conv:
    """One nn.Conv2d repacked for ctk_conv2d_sh: W'[n_pad][(ky*KW + kx)*Cin + c], split-half packed."""

    def __init__(self, conv: torch.nn.Conv2d, device, stem=False):
        w = conv.weight.detach().to(device=device, dtype=torch.float32)
        n, cin, kh, kw = w.shape
        self.n_out, self.kh, self.kw = n, kh, kw
        self.stride, self.pad = conv.stride[0], conv.padding[0]
        self.n_pad = (n + 127) // 128 * 128
        m = w.permute(0, 2, 3, 1).reshape(n, kh * kw * cin)          # [n][(ky,kx,c)]
        if stem:                                                     # the 7x7x3 patch, 147 -> 160 columns: a 1x1 convolution on the im2col rows
            m = torch.nn.functional.pad(m, (0, 160 - m.shape[1]))
            self.cin, self.kh, self.kw, self.stride, self.pad = 160, 1, 1, 1, 0
        else:
            assert cin % 32 == 0
            self.cin = cin
        mp = torch.zeros(self.n_pad, m.shape[1], device=device, dtype=torch.float32)
        mp[:n] = m
        self.wp = ops.pack_weight(mp.contiguous())
        self.bias = torch.zeros(self.n_pad, device=device, dtype=torch.float32)
        self.bias[:n] = conv.bias.detach().to(device=device, dtype=torch.float32)


class HipEncoder:
    def __init__(self, fnet, device, normalize=True):
        self.device = device
        self.normalize = bool(normalize)  # CoTracker3 L2-normalises the features (cotracker3_online.py:373-376); CoTracker2 does not
        self.stride = fnet.stride
        self.zeros = torch.zeros(64, device=device, dtype=torch.float32)
        self.conv1 = _Conv(fnet.conv1, device, stem=True)
        self.units = []
        for layer in (fnet.layer1, fnet.layer2, fnet.layer3, fnet.layer4):
            us = []
            for u in layer:
                us.append((_Conv(u.conv1, device), _Conv(u.conv2, device),
                           _Conv(u.downsample[0], device) if u.downsample is not None else None))
            self.units.append(us)
        self.conv2 = _Conv(fnet.conv2, device)
        self.conv3 = _Conv(fnet.conv3, device)
        self.trace = None  # dev: a dict that receives named f32 intermediates (tools/check_encoder_hip.py)

    # ---- primitives -------------------------------------------------------------------------------------------
    def _conv(self, x_sh, F, H, W, cv: _Conv):
        Ho = (H + 2 * cv.pad - cv.kh) // cv.stride + 1
        Wo = (W + 2 * cv.pad - cv.kw) // cv.stride + 1
        out = torch.empty(F * Ho * Wo, cv.n_out, device=self.device, dtype=torch.float32)
        L.check(L.load().ctk_conv2d_sh(_ptr(x_sh), F, H, W, cv.cin, _ptr(cv.wp), _ptr(cv.bias), cv.n_out, cv.n_pad, cv.kh, cv.kw,
                                       cv.stride, cv.pad, _ptr(out), _ptr(self.zeros), ops._stream()), "ctk_conv2d_sh")
        return out, Ho, Wo

    def _stats(self, x, F, HW, Cn):
        nb = C.c_size_t()
        L.check(L.load().ctk_enc_inorm_workspace_bytes(F, HW, Cn, C.byref(nb)), "ctk_enc_inorm_workspace_bytes")
        ws = torch.empty(nb.value // 8, device=self.device, dtype=torch.float64)
        st = torch.empty(F, Cn, 2, device=self.device, dtype=torch.float32)
        L.check(L.load().ctk_enc_inorm_stats(_ptr(x), F, HW, Cn, 1e-5, _ptr(st), _ptr(ws), ops._stream()), "ctk_enc_inorm_stats")
        return st

    def _apply(self, x, st, F, HW, Cn, skip=None, skip_st=None, want_sh=True, want_f32=False):
        sh = torch.empty(F * HW, Cn // 32, 2, 32, device=self.device, dtype=torch.float16) if want_sh else None
        f32 = torch.empty(F * HW, Cn, device=self.device, dtype=torch.float32) if want_f32 else None
        L.check(L.load().ctk_enc_inorm_apply(_ptr(x), _ptr(st), _ptr(skip), _ptr(skip_st), F, HW, Cn, _ptr(sh), _ptr(f32), ops._stream()),
                "ctk_enc_inorm_apply")
        return sh, f32

    def _unit(self, x_sh, x_f32, F, H, W, unit):
        c1, c2, cd = unit
        y, Ho, Wo = self._conv(x_sh, F, H, W, c1)
        y_sh, _ = self._apply(y, self._stats(y, F, Ho * Wo, c1.n_out), F, Ho * Wo, c1.n_out)
        z, _, _ = self._conv(y_sh, F, Ho, Wo, c2)
        zst = self._stats(z, F, Ho * Wo, c2.n_out)
        if cd is not None:
            d, _, _ = self._conv(x_sh, F, H, W, cd)
            out_sh, out = self._apply(z, zst, F, Ho * Wo, c2.n_out, skip=d, skip_st=self._stats(d, F, Ho * Wo, cd.n_out), want_f32=True)
        else:
            out_sh, out = self._apply(z, zst, F, Ho * Wo, c2.n_out, skip=x_f32, want_f32=True)
        return out_sh, out, Ho, Wo

    # ---- forward ----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, frames: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        """frames [F,3,H,W] float32 in 0..255 (the 2*(x/255)-1 of cotracker3_online.py:320 happens in the stem kernel)
        -> L2-normalised NHWC features [F, H/stride, W/stride, 128]."""
        assert frames.is_cuda and frames.dtype == torch.float32 and frames.is_contiguous() and frames.shape[1] == 3
        F, _, H, W = frames.shape
        lib = L.load()
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        stem = torch.empty(F * Ho * Wo, 5, 2, 32, device=self.device, dtype=torch.float16)
        L.check(lib.ctk_enc_stem_im2col(_ptr(frames), F, H, W, _ptr(stem), ops._stream()), "ctk_enc_stem_im2col")
        x, _, _ = self._conv(stem, F, Ho, Wo, self.conv1)
        del stem
        if self.trace is not None:
            self.trace["conv1"] = x.view(F, Ho, Wo, 64)
        x_sh, x_f32 = self._apply(x, self._stats(x, F, Ho * Wo, 64), F, Ho * Wo, 64, want_f32=True)
        h, w = Ho, Wo
        stages = []
        for us in self.units:
            for u in us:
                x_sh, x_f32, h, w = self._unit(x_sh, x_f32, F, h, w, u)
            stages.append((x_f32, h, w, x_f32.shape[1]))
            if self.trace is not None:
                self.trace[f"layer{len(stages)}"] = x_f32.view(F, h, w, -1)
        oh, ow = H // self.stride, W // self.stride
        ctot = sum(s[3] for s in stages)
        fused = torch.empty(F * oh * ow, ctot // 32, 2, 32, device=self.device, dtype=torch.float16)
        srcs = (C.c_void_p * 4)(*[s[0].data_ptr() for s in stages])
        hs = (C.c_int32 * 4)(*[s[1] for s in stages])
        ws_ = (C.c_int32 * 4)(*[s[2] for s in stages])
        cs = (C.c_int32 * 4)(*[s[3] for s in stages])
        L.check(lib.ctk_enc_fuse(srcs, hs, ws_, cs, F, oh, ow, _ptr(fused), ops._stream()), "ctk_enc_fuse")
        y, _, _ = self._conv(fused, F, oh, ow, self.conv2)
        if self.trace is not None:
            self.trace["fused"] = ops.unsplit(fused).view(F, oh, ow, ctot)
            self.trace["conv2"] = y.view(F, oh, ow, 256)
        y_sh, _ = self._apply(y, self._stats(y, F, oh * ow, 256), F, oh * ow, 256)
        z, _, _ = self._conv(y_sh, F, oh, ow, self.conv3)
        if self.trace is not None:
            self.trace["conv3"] = z.view(F, oh, ow, 128)
        if out is None:
            out = torch.empty(F, oh, ow, 128, device=self.device, dtype=torch.float32)
        assert out.shape == (F, oh, ow, 128) and out.is_contiguous()
        if self.normalize:
            L.check(lib.ctk_enc_l2norm(_ptr(z), F * oh * ow, _ptr(out), ops._stream()), "ctk_enc_l2norm")
        else:
            out.view(F * oh * ow, 128).copy_(z)
        return out
