"""Synthetic-weight waifu2x graphs (cunet / upcunet / swin_unet) and their ONNX export.

The reference consumes ONNX files that are release assets and are NOT in its tree
(/root/reference/README.md:11-15, path built at src/main.cpp:201-204); they derive from
nagadomi/nunif (README.md:99).  No .onnx exists offline, so the architectures are
restated here as plain torch modules (our own code, written from the published
architecture: valid 3x3 convs + LeakyReLU(0.1), SE blocks, 2x2/s2 down, 2x2/s2 and
4x4/s2/p3 transposed convs for cunet; conv stem + shifted-window attention blocks at
three resolutions + linear/pixel-shuffle heads for swin_unet) with seeded random
weights, and exported with torch's TorchScript ONNX exporter.  These files are the
"same ONNX weights" both the oracle and the HIP engine read.

Geometry (what the tile grid depends on, img2img_render.cpp:16-19):
  upcunet (cunet/art scale2):  T' = 2T - 72
  cunet   (cunet/art scale1):  T' = T - 56
  swin_unet scale s:           T' = s (T - 16),  (T-16) % 24 == 0
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- cunet
class SEBlock(nn.Module):
    def __init__(self, ch, reduction=8):
        super().__init__()
        self.conv1 = nn.Conv2d(ch, ch // reduction, 1, 1, 0)
        self.conv2 = nn.Conv2d(ch // reduction, ch, 1, 1, 0)

    def forward(self, x):
        s = torch.mean(x, dim=(2, 3), keepdim=True)
        s = F.relu(self.conv1(s))
        s = torch.sigmoid(self.conv2(s))
        return x * s


class UNetConv(nn.Module):
    def __init__(self, cin, cmid, cout, se):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(cin, cmid, 3, 1, 0), nn.LeakyReLU(0.1),
            nn.Conv2d(cmid, cout, 3, 1, 0), nn.LeakyReLU(0.1))
        self.seblock = SEBlock(cout, 8) if se else None

    def forward(self, x):
        z = self.conv(x)
        if self.seblock is not None:
            z = self.seblock(z)
        return z


class UNet1(nn.Module):
    def __init__(self, cin, cout, deconv):
        super().__init__()
        self.conv1 = UNetConv(cin, 32, 64, se=False)
        self.conv1_down = nn.Conv2d(64, 64, 2, 2, 0)
        self.conv2 = UNetConv(64, 128, 64, se=True)
        self.conv2_up = nn.ConvTranspose2d(64, 64, 2, 2, 0)
        self.conv3 = nn.Conv2d(64, 64, 3, 1, 0)
        if deconv:
            self.conv_bottom = nn.ConvTranspose2d(64, cout, 4, 2, 3)
        else:
            self.conv_bottom = nn.Conv2d(64, cout, 3, 1, 0)

    def forward(self, x):
        x1 = self.conv1(x)
        x2 = F.leaky_relu(self.conv1_down(x1), 0.1)
        x2 = self.conv2(x2)
        x2 = F.leaky_relu(self.conv2_up(x2), 0.1)
        x1 = F.pad(x1, (-4, -4, -4, -4))
        x3 = F.leaky_relu(self.conv3(x1 + x2), 0.1)
        return self.conv_bottom(x3)


class UNet2(nn.Module):
    def __init__(self, cin, cout, deconv):
        super().__init__()
        self.conv1 = UNetConv(cin, 32, 64, se=False)
        self.conv1_down = nn.Conv2d(64, 64, 2, 2, 0)
        self.conv2 = UNetConv(64, 64, 128, se=True)
        self.conv2_down = nn.Conv2d(128, 128, 2, 2, 0)
        self.conv3 = UNetConv(128, 256, 128, se=True)
        self.conv3_up = nn.ConvTranspose2d(128, 128, 2, 2, 0)
        self.conv4 = UNetConv(128, 64, 64, se=True)
        self.conv4_up = nn.ConvTranspose2d(64, 64, 2, 2, 0)
        self.conv5 = nn.Conv2d(64, 64, 3, 1, 0)
        if deconv:
            self.conv_bottom = nn.ConvTranspose2d(64, cout, 4, 2, 3)
        else:
            self.conv_bottom = nn.Conv2d(64, cout, 3, 1, 0)

    def forward(self, x):
        x1 = self.conv1(x)
        x2 = F.leaky_relu(self.conv1_down(x1), 0.1)
        x2 = self.conv2(x2)
        x3 = F.leaky_relu(self.conv2_down(x2), 0.1)
        x3 = self.conv3(x3)
        x3 = F.leaky_relu(self.conv3_up(x3), 0.1)
        x2 = F.pad(x2, (-4, -4, -4, -4))
        x4 = self.conv4(x2 + x3)
        x4 = F.leaky_relu(self.conv4_up(x4), 0.1)
        x1 = F.pad(x1, (-16, -16, -16, -16))
        x5 = F.leaky_relu(self.conv5(x1 + x4), 0.1)
        return self.conv_bottom(x5)


class CUNet(nn.Module):
    """scale 1 (noise only): T' = T - 56."""
    scale = 1

    def __init__(self, cin=3, cout=3):
        super().__init__()
        self.unet1 = UNet1(cin, cout, deconv=False)
        self.unet2 = UNet2(cin, cout, deconv=False)

    def forward(self, x):
        x = self.unet1(x)
        x0 = self.unet2(x)
        x1 = F.pad(x, (-20, -20, -20, -20))
        return torch.clamp(x0 + x1, 0.0, 1.0)


class UpCUNet(nn.Module):
    """scale 2: T' = 2T - 72."""
    scale = 2

    def __init__(self, cin=3, cout=3):
        super().__init__()
        self.unet1 = UNet1(cin, cout, deconv=True)
        self.unet2 = UNet2(cin, cout, deconv=False)

    def forward(self, x):
        x = self.unet1(x)
        x0 = self.unet2(x)
        x1 = F.pad(x, (-20, -20, -20, -20))
        return torch.clamp(x0 + x1, 0.0, 1.0)


# ----------------------------------------------------------------------------- swin_unet
def _rel_pos_index(ws):
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij"))
    flat = coords.flatten(1)
    rel = flat[:, :, None] - flat[:, None, :]
    rel = rel.permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1).flatten()


def _shift_mask(H, W, ws, shift):
    """The shifted-window attention mask [nW, ws*ws, ws*ws] with -100 / 0 entries."""
    m = torch.zeros(H, W)
    cnt = 0
    for hs in ((0, -ws), (-ws, -shift), (-shift, None)):
        for wsl in ((0, -ws), (-ws, -shift), (-shift, None)):
            m[hs[0]:hs[1], wsl[0]:wsl[1]] = cnt
            cnt += 1
    m = m.view(H // ws, ws, W // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    d = m.unsqueeze(1) - m.unsqueeze(2)
    return torch.where(d != 0, torch.full_like(d, -100.0), torch.zeros_like(d))


class SwinBlock(nn.Module):
    """Swin-V1 block on BHWC maps: x += proj(W-MSA(LN(x))); x += MLP(LN(x)), MLP ratio 2."""

    def __init__(self, dim, heads, ws, shift, tv=False):
        super().__init__()
        self.dim, self.heads, self.ws, self.shift = dim, heads, ws, shift
        self.tv = tv        # trace the attention the way torchvision.models.swin_transformer.shifted_window_attention writes it (attn_tv)
        self.norm1 = nn.LayerNorm(dim)
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)
        self.rpb_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
        self.register_buffer("rpb_index", _rel_pos_index(ws), persistent=False)
        self.norm2 = nn.LayerNorm(dim)
        self.fc1 = nn.Linear(dim, dim * 2)
        self.fc2 = nn.Linear(dim * 2, dim)

    def attn(self, x):
        B, H, W, C = x.shape
        ws, sh, nh = self.ws, self.shift, self.heads
        if sh > 0:
            x = torch.roll(x, shifts=(-sh, -sh), dims=(1, 2))
        nW = (H // ws) * (W // ws)
        x = x.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B * nW, ws * ws, C)
        qkv = self.qkv(x).reshape(B * nW, ws * ws, 3, nh, C // nh).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        q = q * (C // nh) ** -0.5
        a = q.matmul(k.transpose(-2, -1))
        bias = self.rpb_table[self.rpb_index].view(ws * ws, ws * ws, nh).permute(2, 0, 1).unsqueeze(0)
        a = a + bias
        if sh > 0:
            mask = _shift_mask(H, W, ws, sh).to(a.dtype)
            a = a.view(B, nW, nh, ws * ws, ws * ws) + mask.unsqueeze(1).unsqueeze(0)
            a = a.view(B * nW, nh, ws * ws, ws * ws)
        a = F.softmax(a, dim=-1)
        x = a.matmul(v).transpose(1, 2).reshape(B * nW, ws * ws, C)
        x = self.proj(x)
        x = x.view(B, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)
        if sh > 0:
            x = torch.roll(x, shifts=(sh, sh), dims=(1, 2))
        return x

    def attn_tv(self, x):
        """The same attention in the operator order of torchvision's shifted_window_attention (the function nunif's swin_unet blocks most likely
        trace through; torchvision is not installed here, this follows its source from memory): a zero F.pad to the window multiple in front (a Pad node
        with all-zero pads when the map already is one), the shift mask built INSIDE the traced function - new_zeros, slice assignments of the region
        ids, view / permute, a difference of two unsqueezes, two masked_fill - and a slice back to [:, :H, :W, :] behind the reverse roll."""
        B, H, W, C = x.shape
        ws, nh = self.ws, self.heads
        pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
        x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
        _, pad_H, pad_W, _ = x.shape
        sh = [self.shift, self.shift]
        if ws >= pad_H: sh[0] = 0
        if ws >= pad_W: sh[1] = 0
        if sum(sh) > 0:
            x = torch.roll(x, shifts=(-sh[0], -sh[1]), dims=(1, 2))
        nW = (pad_H // ws) * (pad_W // ws)
        x = x.view(B, pad_H // ws, ws, pad_W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B * nW, ws * ws, C)
        qkv = F.linear(x, self.qkv.weight, self.qkv.bias)
        qkv = qkv.reshape(x.size(0), x.size(1), 3, nh, C // nh).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        q = q * (C // nh) ** -0.5
        attn = q.matmul(k.transpose(-2, -1))
        N = ws * ws
        bias = self.rpb_table[self.rpb_index].view(N, N, -1).permute(2, 0, 1).contiguous().unsqueeze(0)
        attn = attn + bias
        if sum(sh) > 0:
            attn_mask = x.new_zeros((pad_H, pad_W))
            h_slices = ((0, -ws), (-ws, -sh[0]), (-sh[0], None))
            w_slices = ((0, -ws), (-ws, -sh[1]), (-sh[1], None))
            count = 0
            for h in h_slices:
                for w in w_slices:
                    attn_mask[h[0]:h[1], w[0]:w[1]] = count
                    count += 1
            attn_mask = attn_mask.view(pad_H // ws, ws, pad_W // ws, ws)
            attn_mask = attn_mask.permute(0, 2, 1, 3).reshape(nW, ws * ws)
            attn_mask = attn_mask.unsqueeze(1) - attn_mask.unsqueeze(2)
            attn_mask = attn_mask.masked_fill(attn_mask != 0, float(-100.0)).masked_fill(attn_mask == 0, float(0.0))
            attn = attn.view(x.size(0) // nW, nW, nh, x.size(1), x.size(1))
            attn = attn + attn_mask.unsqueeze(1).unsqueeze(0)
            attn = attn.view(-1, nh, x.size(1), x.size(1))
        attn = F.softmax(attn, dim=-1)
        x = attn.matmul(v).transpose(1, 2).reshape(x.size(0), x.size(1), C)
        x = F.linear(x, self.proj.weight, self.proj.bias)
        x = x.view(B, pad_H // ws, pad_W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, pad_H, pad_W, C)
        if sum(sh) > 0:
            x = torch.roll(x, shifts=(sh[0], sh[1]), dims=(1, 2))
        return x[:, :H, :W, :].contiguous()

    def forward(self, x):
        x = x + (self.attn_tv if self.tv else self.attn)(self.norm1(x))
        x = x + self.fc2(F.gelu(self.fc1(self.norm2(x))))
        return x


class SwinBlocks(nn.Module):
    def __init__(self, dim, heads, layers, ws, tv=False):
        super().__init__()
        self.block = nn.Sequential(*[
            SwinBlock(dim, heads, ws, 0 if i % 2 == 0 else ws // 2, tv) for i in range(layers)])

    def forward(self, x):
        return self.block(x)


class PatchDown(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 2, 2, 0)

    def forward(self, x):  # BHWC -> BHWC
        x = self.conv(x.permute(0, 3, 1, 2))
        return x.permute(0, 2, 3, 1)


class PatchUp(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.proj = nn.Linear(cin, cout * 4)

    def forward(self, x):  # BHWC -> B(2H)(2W)C
        x = self.proj(x).permute(0, 3, 1, 2)
        x = F.pixel_shuffle(x, 2)
        return x.permute(0, 2, 3, 1)


class ToImage(nn.Module):
    def __init__(self, cin, cout, scale):
        super().__init__()
        self.scale = scale
        self.proj = nn.Linear(cin, cout * scale * scale)

    def forward(self, x):  # BHWC -> BCHW (upscaled)
        x = self.proj(x).permute(0, 3, 1, 2)
        if self.scale > 1:
            x = F.pixel_shuffle(x, self.scale)
        return x


class SwinUNet(nn.Module):
    """swin_unet/{art,art_scan,photo}: T' = scale (T - 16)."""

    def __init__(self, cin=3, cout=3, base_dim=96, base_layers=2, scale=4, ws=6, heads=None, tv=False):
        super().__init__()
        C, Hd, L = base_dim, heads or base_dim // 16, base_layers
        self.scale = scale
        self.patch = nn.Sequential(
            nn.Conv2d(cin, C // 2, 3, 1, 0), nn.LeakyReLU(0.1),
            nn.Conv2d(C // 2, C, 3, 1, 0), nn.LeakyReLU(0.1))
        self.swin1 = SwinBlocks(C, Hd, L, ws, tv)
        self.down1 = PatchDown(C, C * 2)
        self.swin2 = SwinBlocks(C * 2, Hd, L, ws, tv)
        self.down2 = PatchDown(C * 2, C * 2)
        self.swin3 = SwinBlocks(C * 2, Hd, L * 3, ws, tv)
        self.up2 = PatchUp(C * 2, C * 2)
        self.swin4 = SwinBlocks(C * 2, Hd, L, ws, tv)
        self.up1 = PatchUp(C * 2, C)
        self.swin5 = SwinBlocks(C, Hd, L, ws, tv)
        self.to_image = ToImage(C, cout, scale)

    def forward(self, x):
        x2 = self.patch(x)
        x2 = F.pad(x2, (-6, -6, -6, -6))
        x2 = x2.permute(0, 2, 3, 1)
        x3 = self.swin1(x2)
        x4 = self.swin2(self.down1(x3))
        x5 = self.swin3(self.down2(x4))
        x5 = self.up2(x5)
        x = self.swin4(x5 + x4)
        x = self.up1(x) + x3
        x = self.swin5(x)
        return torch.clamp(self.to_image(x), 0.0, 1.0)


# ----------------------------------------------------------------------------- factory / init
def output_tile_size(model: str, scale: int, tile: int) -> int:
    if model.startswith("cunet"):
        return 2 * tile - 72 if scale == 2 else tile - 56
    return scale * (tile - 16)


def make_model(model: str, scale: int, seed: int = 1234, small: bool = False, variant: dict | None = None) -> nn.Module:
    """model in {cunet/art, swin_unet/art, swin_unet/art_scan, swin_unet/photo} (main.cpp:26-33).
    variant (swin_unet only): other transformer shapes than the release graphs' - {"ws": window, "heads": heads of the first level,
    "base_dim": channels, "tv": 1 = the attention traced in torchvision's shifted_window_attention operator order} - for the loader's robustness
    tests (graphs the builder did not write the kernels for)."""
    torch.manual_seed(seed)
    if model.startswith("cunet"):
        if scale == 4:
            raise ValueError("cunet/art has no scale 4 (main.cpp:142-143)")
        net = UpCUNet() if scale == 2 else CUNet()
    elif model.startswith("swin_unet"):
        v = dict(variant or {})
        net = SwinUNet(scale=scale, base_dim=v.get("base_dim", 48 if small else 96), ws=v.get("ws", 6), heads=v.get("heads"), tv=bool(v.get("tv", 0)))
    else:
        raise ValueError(model)
    _init(net, seed)
    return net.eval()


@torch.no_grad()
def _init(net: nn.Module, seed: int):
    """Seeded N(0, 1/fan_in)-style weights scaled so activations stay O(1) through the depth
    and the output lands inside (0,1) for most pixels (so Clip and the u8 rounding are both
    exercised without saturating the whole frame)."""
    g = torch.Generator().manual_seed(seed)
    for name, m in net.named_modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            if isinstance(m, nn.ConvTranspose2d):
                fan_in = m.in_channels * m.kernel_size[0] * m.kernel_size[1] / (m.stride[0] * m.stride[1])
            else:
                fan_in = m.in_channels * m.kernel_size[0] * m.kernel_size[1]
            m.weight.copy_(torch.randn(m.weight.shape, generator=g) * math.sqrt(1.6 / fan_in))
            m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
        elif isinstance(m, nn.Linear):
            m.weight.copy_(torch.randn(m.weight.shape, generator=g) * math.sqrt(1.0 / m.in_features))
            m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
        elif isinstance(m, nn.LayerNorm):
            m.weight.copy_(1.0 + 0.1 * torch.randn(m.weight.shape, generator=g))
            m.bias.copy_(0.05 * torch.randn(m.bias.shape, generator=g))
        if isinstance(m, SwinBlock):
            m.rpb_table.copy_(torch.randn(m.rpb_table.shape, generator=g) * 0.5)
            # keep the residual stream bounded: branch outputs are damped
            m.proj.weight.mul_(0.5)
            m.fc2.weight.mul_(0.5)
    # heads: small weights around a mid-grey bias so outputs sit inside (0,1)
    if isinstance(net, SwinUNet):
        net.to_image.proj.weight.mul_(0.025)
        net.to_image.proj.bias.fill_(0.5)
    else:
        for u in (net.unet1, net.unet2):
            u.conv_bottom.weight.mul_(0.15)
        net.unet1.conv_bottom.bias.fill_(0.5)
        net.unet2.conv_bottom.bias.fill_(0.0)


def export_onnx(net: nn.Module, path: str, batch: int, tile: int, opset: int = 17,
                dynamic: bool = False, script: bool = False, **exporter_kw) -> str:
    """TorchScript exporter; the onnx python package is absent, so its no-op
    post-processing hook is bypassed.  exporter_kw: the exporter's own switches as torch.onnx.export takes them
    (do_constant_folding, keep_initializers_as_inputs, training, dynamic_axes ...); script: export torch.jit.script(net) instead of tracing it
    (tests/test_loader_exporter_switches.py)."""
    import torch.onnx._internal.torchscript_exporter.onnx_proto_utils as opu
    opu._add_onnxscript_fn = lambda model_bytes, custom_opsets: model_bytes
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    x = torch.zeros(batch, 3, tile, tile)
    kw = {}
    if dynamic:
        kw["dynamic_axes"] = {"x": {0: "b"}, "y": {0: "b"}}
    # tracing must not record an autograd graph: with trainable parameters every intermediate of the example run stays
    # alive until the end, which at batch 16 / tile 640 is hundreds of GB of host memory
    net.eval()
    for prm in net.parameters():
        prm.requires_grad_(False)
    import warnings
    kw.update(exporter_kw)
    kw.setdefault("do_constant_folding", True)
    with warnings.catch_warnings(), torch.no_grad():
        warnings.simplefilter("ignore")
        torch.onnx.export(torch.jit.script(net) if script else net, (x,), path, dynamo=False, opset_version=opset,
                          input_names=["x"], output_names=["y"], **kw)
    return path


def model_path(root: str, model: str, scale: int, noise: int) -> str:
    """models/<model>/[noiseN_][scaleSx].onnx  (main.cpp:201-204, incl. the scale-1 trailing '_')."""
    name = ("" if noise == -1 else f"noise{noise}_") + ("" if scale == 1 else f"scale{scale}x")
    return os.path.join(root, "models", model, name + ".onnx")


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="swin_unet/art")
    ap.add_argument("--scale", type=int, default=4)
    ap.add_argument("--noise", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--root", default=".")
    ap.add_argument("--opset", type=int, default=17)
    a = ap.parse_args()
    p = model_path(a.root, a.model, a.scale, a.noise)
    export_onnx(make_model(a.model, a.scale, seed=1234 + a.noise), p, a.batch, a.tile, a.opset)
    print(p, os.path.getsize(p))
