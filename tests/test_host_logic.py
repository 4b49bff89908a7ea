"""Host-side logic on CPU: reference-compatible state_dict schemas, the grouped<->per-part key
mapping, the portable synthetic generator, flat parameter buffers, batch sharding."""
import json
import os

import numpy as np
import pytest
import torch

from jafpro_amd import synth


def _schema(golden_dir):
    return json.load(open(os.path.join(golden_dir, "state_dict_schema.json")))


def _modules():
    from jafpro_amd.convLSTM import ConvLSTM
    from jafpro_amd.crn_model import CRN_smaller
    from jafpro_amd.flow_net import Propagation3DFlowNet
    from jafpro_amd.networks import Accumulate_LSTM_no_loss, FaceDiscriminator, ImageDiscriminator, UNet_inpainter
    return {"Accumulate_LSTM_no_loss": Accumulate_LSTM_no_loss, "UNet_inpainter": UNet_inpainter,
            "CRN_smaller_fg": lambda: CRN_smaller(3, fg=True), "CRN_smaller": lambda: CRN_smaller(3),
            "Propagation3DFlowNet": lambda: Propagation3DFlowNet(9, 32, 2, 3, use_deconv=False),
            "ImageDiscriminator": lambda: ImageDiscriminator(32, 6), "FaceDiscriminator": lambda: FaceDiscriminator(32, 6),
            "ConvLSTM": lambda: ConvLSTM((7, 5), 4, [4], [(3, 3)], 1, batch_first=True, bias=True)}


@pytest.mark.parametrize("name", ["Accumulate_LSTM_no_loss", "UNet_inpainter", "CRN_smaller_fg", "CRN_smaller",
                                  "Propagation3DFlowNet", "ImageDiscriminator", "FaceDiscriminator", "ConvLSTM"])
def test_state_dict_matches_reference_schema(golden_dir, name):
    """Keys, order and shapes recorded from the reference modules (oracle/make_golden.py g_schema)."""
    ref = _schema(golden_dir)[name]
    sd = _modules()[name]().state_dict()
    assert [k for k, _ in ref] == list(sd.keys())
    for k, shape in ref:
        assert list(sd[k].shape) == shape, k


def test_grouped_load_is_the_inverse_of_save():
    from jafpro_amd.networks import Accumulate_LSTM_no_loss
    a = synth.load_synth(Accumulate_LSTM_no_loss(), 3)
    sd = a.state_dict()
    assert len(sd) == 912 and sum(v.numel() for v in sd.values()) == 28392552
    # a per-part key addresses the matching slice of the grouped parameter
    assert torch.equal(sd["Downsampler_list.5.enc3.enconv.0.weight"], a.enc3_w[5 * 24:6 * 24])
    assert torch.equal(sd["Downsampler_list.7.convLSTM2.cell_list.0.conv.bias"], a.lstm2_b[7 * 96:8 * 96])
    b = Accumulate_LSTM_no_loss()
    missing = b.load_state_dict(sd)
    assert not missing.missing_keys and not missing.unexpected_keys
    for (ka, va), (kb, vb) in zip(sd.items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    bad = dict(sd)
    bad.pop("Upsampler_list.3.conv.bias")
    with pytest.raises(RuntimeError):
        Accumulate_LSTM_no_loss().load_state_dict(bad)


def test_synth_is_deterministic_and_key_order_free():
    s1 = synth.synth_state_dict({"a.weight": (4, 3, 3, 3), "b.bias": (4,)}, 7)
    s2 = synth.synth_state_dict({"b.bias": (4,), "a.weight": (4, 3, 3, 3)}, 7)
    assert np.array_equal(s1["a.weight"], s2["a.weight"]) and np.array_equal(s1["b.bias"], s2["b.bias"])
    assert not np.array_equal(s1["a.weight"], synth.synth_state_dict({"a.weight": (4, 3, 3, 3)}, 8)["a.weight"])
    # pinned values: PCG64 streams must not drift between numpy versions / machines
    assert abs(float(synth.uniform(1, "x", (3,))[0]) - float(np.random.default_rng([1, 0x8cdc1683]).uniform(-1, 1, 3).astype(np.float32)[0])) == 0.0


def test_body_mesh_topology():
    v, f = synth.body_mesh()
    assert v.shape == (6890, 3) and f.shape == (13776, 3) and f.min() == 0 and f.max() == 6889
    # closed genus-0 surface: every edge is shared by exactly two faces, F = 2V - 4
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
    _, counts = np.unique(e, axis=0, return_counts=True)
    assert (counts == 2).all() and len(f) == 2 * len(v) - 4


def test_stage4_batch_layouts():
    b = synth.stage4_batch(5, 2)
    assert b["src_texture_im"].shape == (2, 4, 3, 800, 1200) and b["src_mask_im"].shape == (2, 4, 800, 1200)
    assert b["tgt_IUV255"].dtype == np.uint8 and b["tgt_IUV255"].shape == (2, 256, 256, 3)
    I = b["tgt_IUV255"][..., 0]
    assert I.max() == 24 and 0.25 < (I > 0).mean() < 0.45 and set(np.unique(b["src_mask_im"])) == {0.0, 1.0}
    assert b["face_bbox"].shape == (2, 4)


def test_flat_params_alias_module_parameters():
    from jafpro_amd.networks import FaceDiscriminator
    from jafpro_amd.step import FlatParams
    m = synth.load_synth(FaceDiscriminator(32, 6), 2)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    fp = FlatParams(m)
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    p = next(m.parameters())
    assert p.data_ptr() == fp.flat.data_ptr() and p.grad.data_ptr() == fp.grad.data_ptr()
    fp.flat.add_(1.0)
    assert torch.allclose(p, before["main.0.weight"] + 1.0)
    assert all(q.data_ptr() % 16 == 0 for q in m.parameters())


def test_shard_batch():
    from jafpro_amd.dist import shard_batch
    b = synth.stage4_batch(5, 4)
    s0, s1 = shard_batch(b, 0, 2), shard_batch(b, 1, 2)
    assert s0["tgt_img"].shape[0] == 2 and np.array_equal(np.concatenate([s0["tgt_img"], s1["tgt_img"]]), b["tgt_img"])
    with pytest.raises(ValueError):
        shard_batch(b, 0, 3)


def test_limit_hw_queues_respects_the_user_and_an_initialised_gpu(monkeypatch):
    """dist.limit_hw_queues: sets GPU_MAX_HW_QUEUES only while it can still take effect (HIP not initialised) and never
    overrides a value the user exported."""
    import torch
    from jafpro_amd.dist import limit_hw_queues
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "7")
    assert limit_hw_queues() is False and os.environ["GPU_MAX_HW_QUEUES"] == "7"
    monkeypatch.delenv("GPU_MAX_HW_QUEUES")
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: True)
    import pytest
    with pytest.warns(RuntimeWarning, match="after HIP was initialised"):       # too late: said aloud, nothing changed
        assert limit_hw_queues() is False and "GPU_MAX_HW_QUEUES" not in os.environ
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: False)
    assert limit_hw_queues() is True and os.environ["GPU_MAX_HW_QUEUES"] == "2"
    monkeypatch.delenv("GPU_MAX_HW_QUEUES")
    monkeypatch.setenv("JAF_HW_QUEUES", "3")                                    # the pool size is the launcher's to choose
    assert limit_hw_queues() is True and os.environ["GPU_MAX_HW_QUEUES"] == "3"
    monkeypatch.delenv("GPU_MAX_HW_QUEUES")


def test_wgrad_watch_fires_once_after_the_last_weight():
    """ops.watch_wgrads: the callback runs when the LAST watched weight has reported its weight-gradient launches, in whatever
    order they come, exactly once; unwatched weights and a cleared watch do nothing."""
    from jafpro_amd import ops
    a, b, c = (torch.zeros(1) for _ in range(3))
    fired = []
    ops.watch_wgrads([a, b], lambda: fired.append("ab"))
    ops._wgrad_enqueued(c); ops._wgrad_enqueued(b)
    assert fired == []
    ops._wgrad_enqueued(b)                       # (a weight reporting twice does not count twice)
    assert fired == []
    ops._wgrad_enqueued(a)
    assert fired == ["ab"]
    # once the range has been declared complete, another weight-gradient launch into it would race with the message that is
    # already travelling: it raises -- until the watch is cleared at the end of the backward pass
    import pytest
    with pytest.raises(RuntimeError, match="declared complete"):
        ops._wgrad_enqueued(a)
    ops._wgrad_enqueued(c)                       # (an unwatched weight is nobody's business)
    ops.watch_wgrads(None)
    ops._wgrad_enqueued(a); ops._wgrad_enqueued(b)
    assert fired == ["ab"]
    ops.watch_wgrads([a], lambda: fired.append("a"))
    ops.watch_wgrads(None)
    ops._wgrad_enqueued(a)
    assert fired == ["ab"]


def test_flat_params_offsets_match_the_parameter_views():
    """FlatParams.offset(i): element offset of parameter i in the flat buffers (16-byte aligned views) -- what adam_range and the
    multi-rank message split slice by."""
    from jafpro_amd.step import FlatParams
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 5, 3), torch.nn.Conv2d(5, 2, 1))
    f = FlatParams(m)
    esz = f.flat.element_size()
    for i, p in enumerate(f.params):
        assert f.offset(i) % 4 == 0
        assert p.data_ptr() == f.flat.data_ptr() + f.offset(i) * esz
        assert p.grad.data_ptr() == f.grad.data_ptr() + f.offset(i) * esz
    assert f.offset(len(f.params)) == f.flat.numel()


def test_product_has_no_oracle_dependency():
    """The product package must never import the oracle (checked on source text)."""
    root = os.path.join(os.path.dirname(__file__), "..", "jafpro_amd")
    for fn in os.listdir(root):
        if fn.endswith(".py"):
            txt = open(os.path.join(root, fn)).read()
            assert "import oracle" not in txt and "from oracle" not in txt, fn


def test_face_bbox_host_logic_and_transfer_texture_oracle():
    """src/data.py:699-716 on the host (the boxes are host integers in the step): margins, the uint8 wrap of an edge at
    256, the invalid all-zero box; and the TransferTexture restatement on a hand-made case (src/utils.py:369-394)."""
    import numpy as np
    from jafpro_amd import data, synth
    from oracle import data_oracle
    iuv = np.zeros((3, 256, 256, 3), np.uint8)
    iuv[0, 100:120, 50:70, 0] = 23
    iuv[0, 110:130, 60:90, 0] = 24
    iuv[1, 240:256, 250:256, 0] = 24                      # touches the right / bottom border: 256 wraps to 0
    bb = data.face_bbox_from_iuv(iuv)
    assert bb.tolist() == [[48, 92, 98, 132], [248, 0, 238, 0], [0, 0, 0, 0]]
    for i in range(3):
        assert np.array_equal(bb[i], data_oracle.face_bbox(iuv[i]))
    tex = np.zeros((800, 1200, 3), np.uint8)
    tex[200 + 199, 400 + 0] = (7, 8, 9)                   # part 9 = cell (1, 2); U = 255 -> row 199, V = 255 -> column 199-199
    m = np.zeros((4, 4, 3), np.uint8)
    m[1, 2] = (9, 255, 255)
    m[3, 3] = (25, 10, 10)                                # part ids above 24 are background
    out = data_oracle.transfer_texture(tex, m, im=np.full((4, 4, 3), 5, np.uint8))
    assert out[1, 2].tolist() == [7, 8, 9] and out[3, 3].tolist() == [5, 5, 5] and out[0, 0].tolist() == [5, 5, 5]
    raw = synth.stage4_raw(3, 1, S=64)
    b = data_oracle.stage4_batch(raw)
    assert b["src_texture_im"].min() >= -1 and b["src_texture_im"].max() <= 1 and b["smpl_real_mask"].max() == 1.0


def test_uv_map_asset_builders_equal_the_reference(tmp_path, golden_dir):
    """jafpro_amd.mesh (SMPLRenderer's static-UV branch, src/mesh.py:28-77,156-194,368-423,530-568) on regenerated synthetic
    assets against tests/golden/mesh_assets.npz, which oracle/make_golden.py g_mesh made by running the reference's own
    create_uvsampler / create_mapping on the same files: bit for bit, every table."""
    import numpy as np
    from jafpro_amd import mesh, synth
    gold = np.load(os.path.join(golden_dir, "mesh_assets.npz"))
    a = synth.uv_assets(str(tmp_path), seed=int(gold["seed"]))
    assert a["nf"] == int(gold["nf"])
    for T_ in (2, 3, 6):
        got = mesh.create_uvsampler(a["obj"], tex_size=T_)
        assert got.shape == (a["nf"], T_ * T_, 2) and got.dtype == np.float32
        assert np.array_equal(got, gold["uvsampler.%d" % T_]), T_
        assert got.min() >= -1.0 and got.max() <= 1.0
    kw = dict(part_info=a["part_info"], front_info=a["front_info"], head_info=a["head_info"], contain_bg=True)
    n = 0
    for fb in (False, True):
        for name in ("uv", "seg", "uv_seg", "par", "front", "head", "back", "binary"):
            key = "map.%s.%d" % (name, int(fb))
            if key not in gold:
                continue
            got = mesh.create_mapping(name, a["obj"], fill_back=fb, **kw)
            assert got.dtype == gold[key].dtype and np.array_equal(got, gold[key]), key
            assert got.shape[0] == a["nf"] * (2 if fb else 1) + 1        # + the background row
            n += 1
    assert n == 15
    assert np.array_equal(mesh.create_mapping("ids", a["obj"], contain_bg=False), gold["map.ids.nobg"])
    with pytest.raises(ValueError):
        mesh.create_mapping("nope", a["obj"], **kw)


def test_draw_subset_reproduces_the_script_stream():
    """jafpro_amd.train.draw_subset(RandomState(seed)) == the reference script's own draws under np.random.seed(seed)
    (train/4.convLSTM_flowpro_interval.py:249-261: one random(), one choice(4, k, replace=False), and for k > 1 one
    choice(k, 1)), restated here against NumPy's GLOBAL stream, 200 iterations of 10 seeds; all four subset sizes and
    unsorted orders occur; the propagation source is always a drawn reference."""
    import numpy as np
    from jafpro_amd.train import draw_subset

    def script_draw():
        r = np.random.random()
        k = 1 if r < 0.25 else 2 if r < 0.5 else 3 if r < 0.75 else 4
        idx = np.random.choice(4, k, replace=False)
        p = idx[0] if k == 1 else idx[np.random.choice(k, 1)]
        return tuple(int(i) for i in idx), int(np.asarray(p).reshape(-1)[0])

    sizes, unsorted = set(), 0
    for seed in range(10):
        np.random.seed(seed)
        want = [script_draw() for _ in range(200)]
        rng = np.random.RandomState(seed)
        got = [draw_subset(rng) for _ in range(200)]
        assert got == want
        for u, p in got:
            assert p in u and len(set(u)) == len(u)
            sizes.add(len(u))
            unsorted += list(u) != sorted(u)
    assert sizes == {1, 2, 3, 4} and unsorted > 100


def test_run_stage4_loop_bookkeeping_without_a_gpu(tmp_path):
    """The loop's own logic on a stand-in trainer (no kernels): count starts at 12000 (:197) and advances per iteration, iteration
    k is handed iteration k+1's clip and propagation source, the last one none, the checkpoint cadence fires on
    count % interval == 0 and puts every module back to .train() (:515-542), `iters` bounds the run, `epochs=None` cycles."""
    import numpy as np
    import torch
    from jafpro_amd import train

    class _Mods(torch.nn.Module):
        def __init__(self):
            super().__init__()
            for n in ("Accu_model", "inpaint_model", "bg_model", "refine_model", "discriminator", "F_Discriminator", "propagater"):
                setattr(self, n, torch.nn.Linear(2, 2))

    class _Trainer:
        reducer = None

        def __init__(self):
            self.M, self.calls = _Mods(), []
            self.M.bg_model.eval()

        def train_step(self, batch, used, prosrc, next_batch=None, next_prosrc=None):
            self.calls.append((batch["id"], used, prosrc, None if next_batch is None else next_batch["id"], next_prosrc))
            return {"total_loss": torch.zeros(1), "final_output": torch.zeros(1)}

    items = [{"id": i, "src_img": torch.zeros(1, 4, 3, 8, 8), "bg_noise": torch.zeros(1, 3, 8, 8)} for i in range(3)]
    tr = _Trainer()
    hist = train.run_stage4(tr, items, seed=16, ckpt_dir=str(tmp_path), save_interval=2, device="cpu")
    rng = np.random.RandomState(16)
    want = [train.draw_subset(rng) for _ in range(3)]
    assert [(c[1], c[2]) for c in tr.calls] == want
    assert [c[0] for c in tr.calls] == [0, 1, 2] and [c[3] for c in tr.calls] == [1, 2, None]
    assert [c[4] for c in tr.calls] == [want[1][1], want[2][1], None]
    assert [h["count"] for h in hist] == [12001, 12002, 12003]
    import os
    assert sorted(os.listdir(tmp_path)) == sorted("%s_iter_12002.pth" % p for p in ("Accu", "inpaint", "bg", "refine", "D", "FD", "pro"))
    assert tr.M.bg_model.training          # :538, the script's quirk: bg_model.train() after a save
    tr2 = _Trainer()
    hist = train.run_stage4(tr2, items, iters=7, epochs=None, draws=[((0,), 0)] * 100, device="cpu")
    assert [c[0] for c in tr2.calls] == [0, 1, 2, 0, 1, 2, 0] and tr2.calls[-1][3] is None and len(hist) == 7
    tr3 = _Trainer()
    train.run_stage4(tr3, items, draws=[((1, 0), 1), ((2,), 2)], device="cpu")          # the draws run out first
    assert len(tr3.calls) == 2 and tr3.calls[-1][3] is None
    import pytest
    with pytest.raises(ValueError):
        train.run_stage4(_Trainer(), items, draws=[((1, 0), 3)], device="cpu")


def test_environment_switches_are_read_once_at_import():
    """The environment switches the product still reads: JAF_RUN_AHEAD, JAF_RANK_CHECK_EVERY, JAF_DIST_ISSUE_ON_WGRAD,
    JAF_ACCU_SPLIT (jafpro_amd/step.py), JAF_HW_QUEUES (dist.py, covered above), JAFPRO_HIP_LIB (_lib.py).  A child interpreter per
    setting: the values are module constants."""
    import subprocess, sys
    code = ("import jafpro_amd.step as s, jafpro_amd._lib as l; "
            "print(s.RUN_AHEAD, s.RANK_CHECK_EVERY, s.DIST_ISSUE_ON_WGRAD, s.ACCU_SPLIT, l.LIB_PATH.endswith('libjafpro_hip.so'))")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(env):
        e = dict(os.environ)
        for k in ("JAF_RUN_AHEAD", "JAF_RANK_CHECK_EVERY", "JAF_DIST_ISSUE_ON_WGRAD", "JAF_ACCU_SPLIT", "JAFPRO_HIP_LIB"):
            e.pop(k, None)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return r.stdout.split()

    assert run({}) == ["1", "200", "True", "True", "True"]      # (default JAF_RUN_AHEAD=1.5: one and a half steps in flight, step.RUN_AHEAD_HALF)
    assert run({"JAF_RUN_AHEAD": "2"})[0] == "2"
    assert run({"JAF_RUN_AHEAD": "0", "JAF_RANK_CHECK_EVERY": "7", "JAF_DIST_ISSUE_ON_WGRAD": "0", "JAF_ACCU_SPLIT": "0",
                "JAFPRO_HIP_LIB": "/nonexistent/other.so"}) == ["0", "7", "False", "False", "False"]
    # no other JAF_* switch is read anywhere in the product (the A/B hooks of rounds 1-4 are gone)
    import re
    seen = set()
    for dirpath, _, files in os.walk(os.path.join(root, "jafpro_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                seen |= set(re.findall(r"(?:environ(?:\.get)?\(|environ\[|getenv\()\s*\"(JAF\w+)\"", text))
    assert seen == {"JAF_RUN_AHEAD", "JAF_RANK_CHECK_EVERY", "JAF_DIST_ISSUE_ON_WGRAD", "JAF_ACCU_SPLIT", "JAF_HW_QUEUES", "JAFPRO_HIP_LIB"}, seen


def test_division_by_launch_constants_is_exact(tmp_path):
    """csrc/jaf_fdiv.h (the convolution kernels' block-index divisions: multiply-high + shift with a host-made magic number) against
    the C `/` for every divisor up to 70 000, the edge numerators of each and a sweep of small numerators; compiled with gcc."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "t.c"
    src.write_text(r'''
#include <stdio.h>
#include "jaf_fdiv.h"
int main(void) {
    unsigned long bad = 0;
    for (uint32_t d = 1; d <= 70000; ++d) {
        jaf_fdiv f = jaf_fdiv_make(d);
        uint32_t ns[] = {0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, 12345678u, 0x7fffffffu, 0x7ffffffeu, 0x40000000u,
                         (0x7fffffffu / d) * d, (0x7fffffffu / d) * d - 1};
        for (unsigned i = 0; i < sizeof(ns) / sizeof(ns[0]); ++i) { uint32_t n = ns[i] & 0x7fffffffu; if (jaf_fdiv_host(n, f) != n / d) ++bad; }
        if (d < 2000) for (uint32_t n = 0; n < 100000; n += 7) if (jaf_fdiv_host(n, f) != n / d) ++bad;
    }
    uint32_t big[] = {0x7fffffffu, 0x40000001u, 0x3fffffffu, 1000003u, 16777259u};
    for (unsigned i = 0; i < 5; ++i) {
        jaf_fdiv f = jaf_fdiv_make(big[i]);
        for (uint32_t n = 0x7fffff00u; n >= 0x7fffff00u && n <= 0x7fffffffu; ++n) if (jaf_fdiv_host(n, f) != n / big[i]) ++bad;
    }
    printf("%lu\n", bad);
    return bad != 0;
}
''')
    exe = tmp_path / "t"
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jafpro_amd", "csrc")
    subprocess.run(["gcc", "-O2", "-I", csrc, str(src), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "0", r.stdout
