"""the device-side input pipeline against the NumPy restatement of src/data.py:640-773 /
src/utils.py:369-394 (oracle/data_oracle.py): bit-exact (the byte -> float map is evaluated in float64 on both sides)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_stage4_batch_from_uint8_is_bit_exact():
    from jafpro_amd import data, synth
    from oracle import data_oracle
    raw = synth.stage4_raw(700, 2)
    ref = data_oracle.stage4_batch(raw)
    draw = {k: torch.from_numpy(v).cuda() for k, v in raw.items()}
    draw["tgt_IUV_host"] = raw["tgt_IUV_u8"]
    out = data.stage4_batch_from_uint8(draw)
    for k, v in ref.items():
        got = out[k].cpu().numpy() if torch.is_tensor(out[k]) else np.asarray(out[k])
        assert got.shape == v.shape and got.dtype == v.dtype, (k, got.shape, v.shape, got.dtype, v.dtype)
        assert np.array_equal(got, v), k
    assert set(np.unique(ref["src_mask_in_image0"])) == {0.0, 1.0} and 0.2 < ref["src_mask_in_image0"].mean() < 0.6
    assert (ref["face_bbox"][:, 0] != ref["face_bbox"][:, 1]).all()


def test_transfer_texture_and_ragged_shapes():
    from jafpro_amd import data, synth
    from oracle import data_oracle
    raw = synth.stage4_raw(701, 3, S=250)               # S*S % 4 == 0 but rows not 16-byte aligned per image at C=3
    tex = torch.from_numpy(raw["src_texture_u8"][:, 0].copy()).cuda()
    iuv = torch.from_numpy(raw["tgt_IUV_u8"]).cuda()
    im = torch.from_numpy(raw["tgt_img_u8"]).cuda()
    a = data.transfer_texture(tex, iuv).cpu().numpy()
    b = data.transfer_texture(tex[0].contiguous(), iuv, im).cpu().numpy()
    for i in range(3):
        assert np.array_equal(a[i], data_oracle.transfer_texture(raw["src_texture_u8"][i, 0], raw["tgt_IUV_u8"][i]))
        assert np.array_equal(b[i], data_oracle.transfer_texture(raw["src_texture_u8"][0, 0], raw["tgt_IUV_u8"][i], raw["tgt_img_u8"][i]))
    # odd sizes take the scalar tail path
    x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (2, 7, 9, 3), dtype=np.uint8)).cuda()
    ref = ((x.cpu().numpy() / 255.0 - 0.5) * 2).transpose(0, 3, 1, 2).astype(np.float32)
    assert np.array_equal(data.normalize_images(x).cpu().numpy(), ref)
    m = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (5, 11), dtype=np.uint8)).cuda()
    assert np.array_equal(data.normalize_images(m, "unit").cpu().numpy(), (m.cpu().numpy() / 255.0).astype(np.float32))
    # no face pixels -> the invalid all-zero box
    iu = raw["tgt_IUV_u8"].copy()
    iu[0, :, :, 0][np.isin(iu[0, :, :, 0], (23, 24))] = 1
    bb = data.face_bbox_from_iuv(iu)
    assert (bb[0] == 0).all() and np.array_equal(bb[1], data_oracle.face_bbox(iu[1]))


def test_pipeline_feeds_the_train_step():
    """A batch produced on the device trains: same step as from the host-prepared arrays of the oracle."""
    from jafpro_amd import data, synth
    from jafpro_amd.step import _to_dev
    from oracle import data_oracle
    from tests._step_util import LOSSES, build
    raw = synth.stage4_raw(702, 1)
    extra = {k: v for k, v in synth.stage4_batch(702, 1).items()
             if k in ("bg_noise", "tgt_verts", "src_verts", "tgt_cam", "src_cam", "src_verts_refs", "src_cam_refs")}
    draw = {k: torch.from_numpy(v).cuda() for k, v in {**raw, **extra}.items()}
    draw["tgt_IUV_host"] = raw["tgt_IUV_u8"]
    b_dev = data.stage4_batch_from_uint8(draw)
    b_host = _to_dev({**data_oracle.stage4_batch(raw), **extra}, "cuda")
    M, tr, _, _, _, _ = build(1)
    o1 = tr.train_step(b_dev)
    M2, tr2, _, _, _, _ = build(1)
    o2 = tr2.train_step(b_host)
    for k in LOSSES:
        assert abs(float(o1[k].reshape(-1)[0]) - float(o2[k].reshape(-1)[0])) <= 1e-5 * max(1.0, abs(float(o2[k].reshape(-1)[0]))), k
    assert torch.isfinite(o1["final_output"]).all()


def test_transfer_texture_golden(golden_dir):
    """jaf_transfer_texture_u8 vs the outputs of the REFERENCE's TransferTexture (src/utils.py:369-394; fixture made by
    oracle/make_golden.py g_data from the function's own code): bit-exact, batched and shared-atlas forms."""
    import os
    from jafpro_amd import data, synth
    st = dict(np.load(os.path.join(golden_dir, "transfer_texture.npz")))
    raw = synth.stage4_raw(int(st["seed"]), 3)
    tex = torch.from_numpy(raw["src_texture_u8"][:, 0].copy()).cuda()
    iuv = torch.from_numpy(raw["tgt_IUV_u8"]).cuda()
    im = torch.from_numpy(raw["tgt_img_u8"]).cuda()
    a = data.transfer_texture(tex, iuv).cpu().numpy()
    b = data.transfer_texture(tex, iuv, im).cpu().numpy()
    for i in range(3):
        assert np.array_equal(a[i], st["plain.%d" % i]) and np.array_equal(b[i], st["over_image.%d" % i])
    ones = torch.ones((800, 1200, 3), dtype=torch.uint8, device="cuda")
    m = data.transfer_texture(ones, torch.from_numpy(raw["src_IUV0_u8"][:1].copy()).cuda()).cpu().numpy()
    assert np.array_equal(m[0], st["ones_mask.0"])
