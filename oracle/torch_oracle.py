"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by jafpro_amd (the product path); only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.

CPU fp32 restatement (torch.nn.functional on state_dict tensors) of the reference's stage-4
algorithm; every function cites the reference lines it follows.  Pinning: oracle/make_golden.py
imports the reference's own nn.Modules in the survey container, loads the same synthetic
state_dicts, and asserts that these functions reproduce the reference outputs; the resulting
vectors are committed under tests/golden/ (the reference itself cannot travel to the GPU box).
Unpinned parts are listed in DESIGN.md (pretrained VGG weights, trained checkpoints).

All floating-point work here is "a torch fp32 reference of a floating-point kernel"; the
integer/byte side (IUV decode, masks, the rasteriser) is in oracle/raster_oracle.c + NumPy.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
ENC_K = [5, 3, 3, 3, 3, 3, 3, 3, 3]
ENC_S = [1, 2, 1, 2, 1, 2, 1, 2, 1]


# ------------------------------------------------------------------------------------------------
# src/networks.py:868-878 Downsampler, :896-909 Upsampler_SE
# ------------------------------------------------------------------------------------------------
def downsampler(sd: SD, pre: str, x, k=3, s=1):
    return F.leaky_relu(F.conv2d(x, sd[pre + ".enconv.0.weight"], sd[pre + ".enconv.0.bias"], stride=s, padding=k // 2), 0.2)


def upsampler_se(sd: SD, pre: str, x, enc_x, size: int):
    x = F.interpolate(x, size=(size, size), mode="bilinear", align_corners=True)   # nn.UpsamplingBilinear2d (:899)
    x = torch.cat([x, enc_x], 1)
    return F.leaky_relu(F.conv2d(x, sd[pre + ".myconv.0.weight"], sd[pre + ".myconv.0.bias"], padding=1), 0.2)


def encoder9(sd: SD, pre: str, x):
    feats = []
    for i in range(9):
        x = downsampler(sd, "%s.enc%d" % (pre, i + 1), x, ENC_K[i], ENC_S[i])
        feats.append(x)
    return feats


# ------------------------------------------------------------------------------------------------
# src/convLSTM.py:41-56 cell, :102-147 sequence (zero initial state :58-63)
# ------------------------------------------------------------------------------------------------
def convlstm_cell(w, b, x, h, c):
    hidden = w.shape[0] // 4
    cc = F.conv2d(torch.cat((x, h), 1), w, b, padding=1)
    cc_i, cc_f, cc_o, cc_g = torch.split(cc, hidden, dim=1)
    i, f, o, g = torch.sigmoid(cc_i), torch.sigmoid(cc_f), torch.sigmoid(cc_o), torch.tanh(cc_g)
    c_cur = f * c + i * g
    return o * torch.tanh(c_cur), c_cur


def convlstm(w, b, xs: Sequence[torch.Tensor]):
    """xs: list over t of (B,C,H,W) -> (list of h_t, (h_T, c_T))."""
    hidden = w.shape[0] // 4
    h = torch.zeros(xs[0].shape[0], hidden, xs[0].shape[2], xs[0].shape[3])
    c = torch.zeros_like(h)
    hs = []
    for x in xs:
        h, c = convlstm_cell(w, b, x, h, c)
        hs.append(h)
    return hs, (h, c)


# ------------------------------------------------------------------------------------------------
# src/networks.py:1290-1357 Downsampler_convLSTM, :1198-1214 Upsampler_stack_noEmbed,
# :1641-1662 Accumulate_LSTM_no_loss
# ------------------------------------------------------------------------------------------------
def accumulate_part(sd: SD, p: int, xs: Sequence[torch.Tensor]):
    B, T = xs[0].shape[0], len(xs)
    pre = "Downsampler_list.%d" % p
    feats = encoder9(sd, pre, torch.cat(list(xs), 0))                 # T refs on the batch axis (:1317)
    skips = []
    for li, fi in enumerate([0, 2, 4, 6, 8]):
        f = feats[fi]
        seq = [f[B * t:B * (t + 1)] for t in range(T)]
        w = sd["%s.convLSTM%d.cell_list.0.conv.weight" % (pre, li + 1)]
        b = sd["%s.convLSTM%d.cell_list.0.conv.bias" % (pre, li + 1)]
        _, (h, _) = convlstm(w, b, seq)
        skips.append(h)                                                # only last h is consumed (:1346-1355)
    up = "Upsampler_list.%d" % p
    x = skips[4]
    for i, size in enumerate([25, 50, 100, 200]):
        x = upsampler_se(sd, "%s.dec%d" % (up, i + 1), x, skips[3 - i], size)
    return F.conv2d(x, sd[up + ".conv.weight"], sd[up + ".conv.bias"], padding=1)    # no activation (:1205,1213)


def accumulate_forward(sd: SD, x_in) -> List[torch.Tensor]:
    return [accumulate_part(sd, p, x_in[p]) for p in range(24)]


def accumulate_lstm_loss(sd: SD, x_in, src_mask, tgt_mask, tgt_tex):
    """src/networks.py:1607-1639 (stage 1): atlas paste + masked L1 over the targets."""
    parts = accumulate_forward(sd, x_in)
    B = parts[0].shape[0]
    atlas = torch.zeros(B, 3, 800, 1200)
    for i in range(4):
        for j in range(6):
            atlas[:, :, i * 200:(i + 1) * 200, j * 200:(j + 1) * 200] = parts[i * 6 + j]
    common = torch.zeros_like(src_mask[:, 0])
    for i in range(src_mask.shape[1]):
        common = common | src_mask[:, i]
    loss = 0
    for i in range(tgt_mask.shape[1]):
        area = (common & tgt_mask[:, i]).float()
        loss = loss + F.l1_loss(area * atlas, area * tgt_tex[:, 0].float())
    return atlas, loss


# ------------------------------------------------------------------------------------------------
# src/networks.py:1121-1170 Downsampler_stack / Upsampler_stack, :1805-1828 UNet_inpainter
# ------------------------------------------------------------------------------------------------
def inpaint_forward(sd: SD, tex_list) -> List[torch.Tensor]:
    embeds, skips_all = [], []
    for p in range(24):
        pre = "Downsampler_list.%d" % p
        feats = encoder9(sd, pre, tex_list[p])
        embeds.append(downsampler(sd, pre + ".enc_compress", feats[8]))
        skips_all.append([feats[0], feats[2], feats[4], feats[6], feats[8]])
    g = torch.cat(embeds, 1)                                           # 72 x 13 x 13 (:1824)
    outs = []
    for p in range(24):
        up = "Upsampler_list.%d" % p
        sk = skips_all[p]
        x = torch.cat([sk[4], g], 1)                                   # (:1164)
        for i, size in enumerate([25, 50, 100, 200]):
            x = upsampler_se(sd, "%s.dec%d" % (up, i + 1), x, sk[3 - i], size)
        outs.append(F.conv2d(x, sd[up + ".conv.weight"], sd[up + ".conv.bias"], padding=1))
    return outs


# ------------------------------------------------------------------------------------------------
# src/crn_model.py:67-106 LayerNorm / ConvBlock, :243-308 CRN_smaller
# ------------------------------------------------------------------------------------------------
def crn_layernorm(x, gamma, beta, eps=1e-5):
    flat = x.reshape(x.size(0), -1)
    mean = flat.mean(1).view(-1, 1, 1, 1)
    std = flat.std(1).view(-1, 1, 1, 1)                                # Bessel-corrected (:82)
    x = (x - mean) / (std + eps)
    return x * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)


def conv_block(sd: SD, pre: str, x):
    for r in (0, 3):
        x = F.conv2d(x, sd["%s.conv_block.%d.weight" % (pre, r)], sd["%s.conv_block.%d.bias" % (pre, r)], padding=1)
        x = crn_layernorm(x, sd["%s.conv_block.%d.gamma" % (pre, r + 1)], sd["%s.conv_block.%d.beta" % (pre, r + 1)])
        x = F.leaky_relu(x, 0.01)                                      # nn.LeakyReLU() default slope (:100)
    return x


def crn_smaller_forward(sd: SD, label, sp: int, fg: bool):
    pool = lambda t: F.avg_pool2d(t, (3, 3), stride=2, padding=1)
    itp = lambda t, s: F.interpolate(t, s, mode="bilinear", align_corners=True)
    pools = []
    x = label
    for k in range(1, 7):
        x = pool(conv_block(sd, "conv%d_encoder" % k, x))
        pools.append(x)
    net = itp(conv_block(sd, "conv6_decoder", torch.cat([itp(label, sp // 64), pools[5]], 1)), sp // 32)
    for k, div in ((5, 32), (4, 16), (3, 8), (2, 4), (1, 2)):
        inp = torch.cat([itp(label, sp // div), pools[k - 1], net], 1)
        net = itp(conv_block(sd, "conv%d_decoder" % k, inp), sp // (div // 2))
    net = conv_block(sd, "decoder", torch.cat([label, net], 1))
    out = F.conv2d(net, sd["out_conv.weight"], sd["out_conv.bias"])
    if fg:
        return out, torch.sigmoid(F.conv2d(net, sd["fg_conv.weight"], sd["fg_conv.bias"]))
    return out


# ------------------------------------------------------------------------------------------------
# src/flow_net.py:6-141 (ctor args (9,32,2,3,use_deconv=False), train/4...py:146)
# ------------------------------------------------------------------------------------------------
def _bn(sd: SD, pre: str, x, training: bool):
    """nn.BatchNorm2d.forward: momentum 0.1, eps 1e-5; in train mode the running statistics are updated in place and
    num_batches_tracked is incremented (torch/nn/modules/batchnorm.py, as used by src/flow_net.py:13-44 and
    src/networks.py:369-390)."""
    nbt = sd.get(pre + ".num_batches_tracked")
    if training and nbt is not None:
        nbt.add_(1)
    return F.batch_norm(x, sd[pre + ".running_mean"], sd[pre + ".running_var"], sd[pre + ".weight"], sd[pre + ".bias"],
                        training, 0.1, 1e-5)


def _resblock(sd: SD, pre: str, x, training: bool):
    y = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), sd[pre + ".conv_block.1.weight"], sd[pre + ".conv_block.1.bias"])
    y = F.relu(_bn(sd, pre + ".conv_block.2", y, training))
    y = F.conv2d(F.pad(y, (1, 1, 1, 1), mode="reflect"), sd[pre + ".conv_block.5.weight"], sd[pre + ".conv_block.5.bias"])
    return x + _bn(sd, pre + ".conv_block.6", y, training)


def propagation_forward(sd: SD, x: dict, training: bool):
    """BatchNorm running stats in `sd` are updated in place when training."""
    fake, tsf = x["fake_tgt"], x["tsf_image"]
    if x["use_mask"]:
        tsf = tsf * x["tgt_smpl_mask"]
    inp = torch.cat([tsf, fake, x["tgt_IUV"]], 1) if x["use_IUV"] else torch.cat([tsf, fake], 1)
    d = "composite_unet.model_down_img"
    h = F.conv2d(F.pad(inp, (3, 3, 3, 3), mode="reflect"), sd[d + ".1.weight"], sd[d + ".1.bias"])
    h = F.relu(_bn(sd, d + ".2", h, training))
    for ci in (4, 7):
        h = F.conv2d(h, sd["%s.%d.weight" % (d, ci)], sd["%s.%d.bias" % (d, ci)], stride=2, padding=1)
        h = F.relu(_bn(sd, "%s.%d" % (d, ci + 1), h, training))
    h = _resblock(sd, d + ".10", h, training)
    h = _resblock(sd, d + ".11", h, training)
    h = _resblock(sd, "composite_unet.model_res_img.0", h, training)
    u = "composite_unet.model_up_img"
    for ci in (1, 5):
        h = F.interpolate(h, scale_factor=2, mode="bilinear", align_corners=False)   # nn.Upsample(2,'bilinear') (:42)
        h = F.conv2d(h, sd["%s.%d.weight" % (u, ci)], sd["%s.%d.bias" % (u, ci)], padding=1)
        h = F.relu(_bn(sd, "%s.%d" % (u, ci + 1), h, training))
    f = "composite_unet.model_final_w"
    w = torch.sigmoid(F.conv2d(F.pad(h, (3, 3, 3, 3), mode="reflect"), sd[f + ".1.weight"], sd[f + ".1.bias"]))
    return {"pred_target": fake * w + tsf * (1 - w), "weight": w}


# ------------------------------------------------------------------------------------------------
# src/networks.py:356-456 discriminators
# ------------------------------------------------------------------------------------------------
def discriminator_forward(sd: SD, x, training: bool, conv_idx: Sequence[int]):
    for n, ci in enumerate(conv_idx):
        x = F.conv2d(x, sd["main.%d.weight" % ci], None, stride=2, padding=1)
        if n > 0:
            x = _bn(sd, "main.%d" % (ci + 1), x, training)
        x = F.leaky_relu(x, 0.2)
    x = x.reshape(x.size(0), -1)
    x = F.leaky_relu(F.linear(x, sd["classifier.0.weight"], sd["classifier.0.bias"]), 0.2)
    return torch.sigmoid(F.linear(x, sd["classifier.2.weight"], sd["classifier.2.bias"]))


IMAGE_D_CONVS = (0, 2, 5, 8, 11, 14)
FACE_D_CONVS = (0, 2, 5, 8)


# ------------------------------------------------------------------------------------------------
# src/networks.py:70-125 VGG19_CRN / VGGLoss_CRN / vgg_preprocess / VGG_l1_loss
# ------------------------------------------------------------------------------------------------
VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
VGG_TAPS = (2, 7, 12, 21, 30)
VGG_WEIGHTS = [1 / 2.6, 1 / 4.8, 1 / 3.7, 1 / 5.6, 10 / 1.5]


def vgg_preprocess(x):
    x = 255.0 * (x + 1.0) / 2.0
    mean = torch.tensor([103.939, 116.779, 123.68]).view(1, 3, 1, 1)
    return x - mean


def vgg_features(sd: SD, x, pre="vgg_loss.vgg.vgg_model."):
    """Taps are POST-ReLU: torchvision's in-place ReLU aliases the captured tensor;
    AvgPool replaces MaxPool (:76-78); layers after 30 do not reach the loss."""
    feats, idx = [], 0
    for v in VGG_CFG:
        if idx > VGG_TAPS[-1]:
            break
        if v == "M":
            x = F.avg_pool2d(x, 2, 2)
            idx += 1
        else:
            x = F.relu(F.conv2d(x, sd["%s%d.weight" % (pre, idx)], sd["%s%d.bias" % (pre, idx)], padding=1))
            if idx in VGG_TAPS:
                feats.append(x)
            idx += 2
    return feats


def vgg_l1_loss(sd: SD, x, y):
    xp, yp = vgg_preprocess(x), vgg_preprocess(y)
    fx, fy = vgg_features(sd, xp), vgg_features(sd, yp)
    loss = 0
    for w, a, b in zip(VGG_WEIGHTS, fx, fy):
        loss = loss + w * F.l1_loss(a, b.detach())
    return loss + F.l1_loss(xp, yp)


# ------------------------------------------------------------------------------------------------
# train/4.convLSTM_flowpro_interval.py:43-76 texture_warp_pytorch
# ------------------------------------------------------------------------------------------------
def texture_warp(tex_parts: Sequence[torch.Tensor], iuv: np.ndarray, align_corners=False):
    """tex_parts: 24 x (3,200,200); iuv: (S,S,3) uint8 -> (3,S,S).  align_corners: SURVEY F7."""
    IUV = torch.from_numpy(np.ascontiguousarray(iuv))
    U, V = IUV[:, :, 1], IUV[:, :, 2]
    gen = torch.zeros(IUV.size()).unsqueeze(0).permute(0, 3, 1, 2)
    for part in range(1, 25):
        tex = tex_parts[part - 1]
        sel = IUV[:, :, 0] == part
        u = torch.where(sel, U.float(), torch.zeros(U.size()))
        v = torch.where(sel, V.float(), torch.zeros(V.size()))
        x = ((255 - v) / 255. - 0.5) * 2
        y = (u / 255. - 0.5) * 2
        grid = torch.cat([x.unsqueeze(2), y.unsqueeze(2)], dim=2).unsqueeze(0)
        patch = F.grid_sample(tex.unsqueeze(0).float(), grid, mode="bilinear", align_corners=align_corners)
        gen = torch.where(sel, patch, gen)
    return gen.squeeze(0)


# ------------------------------------------------------------------------------------------------
# src/nmr.py:10-28,263-278 projection; third_party/.../look_at.py:6-62; vertices_to_faces.py:4-22
# ------------------------------------------------------------------------------------------------
EYE_Z = -(1. / np.tan(np.radians(30)) + 1)


def look_at(vertices, eye, at=(0, 0, 0), up=(0, 1, 0)):
    at = torch.tensor(at, dtype=torch.float32)
    up = torch.tensor(up, dtype=torch.float32)
    eye = torch.as_tensor(eye, dtype=torch.float32)
    bs = vertices.shape[0]
    eye, at, up = (t[None, :].repeat(bs, 1) if t.ndimension() == 1 else t for t in (eye, at, up))
    z_axis = F.normalize(at - eye, eps=1e-5)
    x_axis = F.normalize(torch.cross(up, z_axis, dim=1), eps=1e-5)
    y_axis = F.normalize(torch.cross(z_axis, x_axis, dim=1), eps=1e-5)
    r = torch.cat((x_axis[:, None, :], y_axis[:, None, :], z_axis[:, None, :]), dim=1)
    return torch.matmul(vertices - eye[:, None, :], r.transpose(1, 2))


def perspective(vertices, angle=30.):
    """third_party/neural_renderer/neural_renderer/perspective.py:6-22 (teapot pin only)."""
    width = torch.tan(torch.tensor(angle / 180 * np.pi, dtype=torch.float32))
    z = vertices[:, :, 2]
    return torch.stack((vertices[:, :, 0] / z / width, vertices[:, :, 1] / z / width, z), dim=2)


def project_faces(verts, cam, faces_idx):
    """verts [B,NV,3], cam [B,3], faces_idx int [NF,3] -> faces [B,NF,3,3] (render_fim_wim up to
    the rasteriser call)."""
    scale = cam[:, 0].contiguous().view(-1, 1, 1)
    trans = cam[:, 1:3].contiguous().view(cam.size(0), 1, -1)
    proj = torch.cat((scale * (verts[:, :, :2] + trans), verts[:, :, 2, None]), 2)
    proj[:, :, 1] *= -1
    v = look_at(proj, [0, 0, EYE_Z])
    return v[:, torch.as_tensor(faces_idx).long()]


def cal_bc_transform(src_f2pts, dst_fims, dst_wims, image_size=256):
    """src/nmr.py:617-659."""
    bs = src_f2pts.shape[0]
    T = -2 * torch.ones((bs, image_size * image_size, 2), dtype=torch.float32)
    for i in range(bs):
        fim = dst_fims[i].long().reshape(-1)
        wim = dst_wims[i].reshape(-1, 3)
        m = fim != -1
        T[i, m] = (src_f2pts[i][fim[m]] * wim[m][:, :, None]).sum(dim=1)
    return T.view(bs, image_size, image_size, 2)


def flow_warp(src_img, src_faces, tgt_fim, tgt_wim, align_corners=False):
    """src/cal_flow.py:28-39 given the two rasterisations."""
    f2 = src_faces[:, :, :, 0:2].clone()
    f2[:, :, :, 1] *= -1
    T = cal_bc_transform(f2, tgt_fim, tgt_wim, tgt_fim.shape[1])
    return F.grid_sample(src_img, T, padding_mode="border", align_corners=align_corners), T


# ------------------------------------------------------------------------------------------------
# train/4.convLSTM_flowpro_interval.py:283-298 common-area masking, :321 fusion blend
# ------------------------------------------------------------------------------------------------
def common_area_mask(src_mask_im, used: Sequence[int]):
    """src_mask_im float [B,4,800,1200]; unused refs are zeroed (:283-286), OR as bytes (:288-292)."""
    m = src_mask_im.clone()
    for i in range(m.shape[1]):
        if i not in used:
            m[:, i] = m[:, i] * 0
    area = torch.zeros_like(m[:, 0]).byte()
    for i in range(m.shape[1]):
        area = area | m[:, i].byte()
    return area.float().unsqueeze(1).repeat(1, 3, 1, 1)


def mask_parts(parts: Sequence[torch.Tensor], area):
    out = []
    for i in range(4):
        for j in range(6):
            out.append(parts[i * 6 + j] * area[:, :, i * 200:(i + 1) * 200, j * 200:(j + 1) * 200])
    return out


# ------------------------------------------------------------------------------------------------
# FlowNetSD (src/flownet2_pytorch/networks/FlowNetSD.py:11-106, submodules.py:7-38) with batchNorm=False, as the
# evaluation script builds it (test/video_evaluation.py:66): functional over the reference's state_dict.
# ------------------------------------------------------------------------------------------------
def flownet_sd_forward(sd: SD, x, training: bool = True):
    lrelu = lambda t: F.leaky_relu(t, 0.1)
    cv = lambda name, t, s=1: lrelu(F.conv2d(t, sd[name + ".0.weight"], sd[name + ".0.bias"], stride=s, padding=1))
    plain = lambda name, t: F.conv2d(t, sd[name + ".weight"], sd[name + ".bias"], stride=1, padding=1)
    icv = lambda name, t: F.conv2d(t, sd[name + ".0.weight"], sd[name + ".0.bias"], stride=1, padding=1)
    dcv = lambda name, t: lrelu(F.conv_transpose2d(t, sd[name + ".0.weight"], sd[name + ".0.bias"], stride=2, padding=1))
    up = lambda name, t: F.conv_transpose2d(t, sd[name + ".weight"], sd[name + ".bias"], stride=2, padding=1)
    c0 = cv("conv0", x)
    c1 = cv("conv1_1", cv("conv1", c0, 2))
    c2 = cv("conv2_1", cv("conv2", c1, 2))
    c3 = cv("conv3_1", cv("conv3", c2, 2))
    c4 = cv("conv4_1", cv("conv4", c3, 2))
    c5 = cv("conv5_1", cv("conv5", c4, 2))
    c6 = cv("conv6_1", cv("conv6", c5, 2))
    flow6 = plain("predict_flow6", c6)
    cat5 = torch.cat((c5, dcv("deconv5", c6), up("upsampled_flow6_to_5", flow6)), 1)
    flow5 = plain("predict_flow5", icv("inter_conv5", cat5))
    cat4 = torch.cat((c4, dcv("deconv4", cat5), up("upsampled_flow5_to_4", flow5)), 1)
    flow4 = plain("predict_flow4", icv("inter_conv4", cat4))
    cat3 = torch.cat((c3, dcv("deconv3", cat4), up("upsampled_flow4_to_3", flow4)), 1)
    flow3 = plain("predict_flow3", icv("inter_conv3", cat3))
    cat2 = torch.cat((c2, dcv("deconv2", cat3), up("upsampled_flow3_to_2", flow3)), 1)
    flow2 = plain("predict_flow2", icv("inter_conv2", cat2))
    return (flow2, flow3, flow4, flow5, flow6) if training else (flow2,)
