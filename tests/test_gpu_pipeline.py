"""GPU: parse glue kernels and the end-to-end try-on step (tocg -> glue -> generator)
against the oracle's composition of test_generator.py:118-219."""
from argparse import Namespace

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import hrviton_oracle as O

pytestmark = pytest.mark.gpu


def _mods():
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import glue, ops
    return glue, ops


def test_resize_nchw_matches_interpolate():
    glue, ops = _mods()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3, 64, 48, generator=g)
    for size in [(16, 12), (32, 24), (50, 31)]:
        want = O.resize_bilinear(x, size=size)
        got = glue.resize_nchw(x.cuda(), size, "bilinear").cpu()
        assert (got - want).abs().max() < 1e-6
        assert torch.equal(glue.resize_nchw(x.cuda(), size, "nearest").cpu(), O.resize_nearest(x, size))


@pytest.mark.parametrize("comp", ["warp_grad", "detach", "no_composition"])
def test_make_parse_vs_oracle(comp):
    glue, ops = _mods()
    g = torch.Generator().manual_seed(1)
    N, h, w, H, W = 2, 32, 24, 128, 96
    seg = F.relu(torch.randn(N, 13, h, w, generator=g))
    cm = torch.rand(N, 1, h, w, generator=g)
    want_g, want_lab, want_parse = O.parse_glue(seg, cm, H, W, comp)
    gauss, labels, parse7 = glue.make_parse(seg.cuda(), cm.cuda(), H, W, comp)
    got_g = ops.to_nchw(gauss).cpu()
    assert (got_g - want_g).abs().max() < 1e-5 * max(1.0, want_g.abs().max().item())
    lab = labels.cpu()[:, 0]
    mism = lab != want_lab
    if mism.any():  # only allowed at near-ties
        top2 = want_g.topk(2, dim=1).values
        assert (top2[:, 0] - top2[:, 1])[mism].max() < 1e-5
    got_parse = ops.to_nchw(parse7).cpu()
    assert got_parse.shape == want_parse.shape
    assert torch.equal(got_parse[~mism[:, None].expand_as(got_parse)], want_parse[~mism[:, None].expand_as(want_parse)])
    assert torch.equal(got_parse.sum(1), torch.ones(N, H, W))


@pytest.mark.parametrize("H,W", [(128, 96), (77, 53)])
def test_gauss_blur_sliding_window_is_bit_identical(monkeypatch, H, W):
    """hrv_gauss_blur_nhwc_f32: eight outputs per thread over a sliding window (the default) == one output per thread (HRV_GAUSS8=0),
    bit for bit -- every output sums the same taps in the same order."""
    glue, ops = _mods()
    from hr_viton_amd import _lib
    g = torch.Generator().manual_seed(3)
    N, h, w = 2, 32, 24
    seg = F.relu(torch.randn(N, 13, h, w, generator=g)).cuda()
    cm = torch.rand(N, 1, h, w, generator=g).cuda()
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("HRV_GAUSS8", flag)
        _lib.reload_env()
        gauss, _, _ = glue.make_parse(seg, cm, H, W, "warp_grad")
        outs.append(gauss.t.clone())
    monkeypatch.delenv("HRV_GAUSS8")
    _lib.reload_env()
    assert torch.equal(outs[0], outs[1])


def test_hires_warp_and_occlusion_vs_oracle():
    glue, ops = _mods()
    g = torch.Generator().manual_seed(2)
    N, H, W = 2, 64, 48
    flow = torch.randn(N, 16, 12, 2, generator=g) * 2.0
    cloth = torch.rand(N, 3, H, W, generator=g) * 2 - 1
    mask = (torch.rand(N, 1, H, W, generator=g) > 0.5).float()
    wc, wm = O.hires_warp(flow, cloth, mask)
    warped = glue.hires_warp(flow.cuda(), cloth.cuda(), mask.cuda())
    assert (ops.to_nchw(warped, 0, 3).cpu() - wc).abs().max() < 2e-5
    assert (ops.to_nchw(warped, 3, 1).cpu() - wm).abs().max() < 2e-5
    gauss = F.relu(torch.randn(N, 13, H, W, generator=g))
    wm2 = O.remove_overlap(F.softmax(gauss, dim=1), wm)
    wc2 = wc * wm2 + torch.ones_like(wc) * (1 - wm2)
    ga = ops.to_nhwc(gauss.cuda())
    glue.occlusion(ga, warped)
    assert (ops.to_nchw(warped, 3, 1).cpu() - wm2).abs().max() < 2e-5
    assert (ops.to_nchw(warped, 0, 3).cpu() - wc2).abs().max() < 5e-5


def test_tryon_step_end_to_end_vs_oracle():
    """tocg(256x192) -> glue -> SPADE generator at 512x384 ('most'), zero noise_scale."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd.network_generator import SPADEGenerator
    from hr_viton_amd.networks import ConditionGenerator
    from hr_viton_amd.pipeline import tryon_step
    H, W = 512, 384
    opt = Namespace(cuda=True, warp_feature="T1", out_layer="relu", norm_G="spectralaliasinstance", gen_semantic_nc=7,
                    ngf=8, num_upsampling_layers="most", fine_height=H, fine_width=W, occlusion=True,
                    clothmask_composition="warp_grad")
    torch.manual_seed(0)
    tocg = ConditionGenerator(opt, 4, 16, 13, ngf=32, norm_layer=nn.BatchNorm2d)
    gen = SPADEGenerator(opt, 9)
    gen.init_weights("xavier", 0.02)
    with torch.no_grad():
        for fc in tocg.flow_conv:
            fc.weight.mul_(4.0)
        for n_, p in gen.named_parameters():
            if n_.endswith("weight") or n_.endswith("weight_orig"):
                p.mul_(30.0)
    tocg.eval()
    gen.eval()
    sd_t = {k: v.detach().clone() for k, v in tocg.state_dict().items()}
    sd_g = {k: v.detach().clone() for k, v in gen.state_dict().items()}
    g = torch.Generator().manual_seed(4)
    N = 1
    lab = torch.randint(0, 13, (N, 1, H // 32, W // 32), generator=g).repeat_interleave(32, 2).repeat_interleave(32, 3)
    inp = {"cloth": torch.rand(N, 3, H, W, generator=g) * 2 - 1, "cloth_mask": (torch.rand(N, 1, H, W, generator=g) > 0.4).float(),
           "parse_agnostic": torch.zeros(N, 13, H, W).scatter_(1, lab, 1.0), "densepose": torch.rand(N, 3, H, W, generator=g) * 2 - 1,
           "agnostic": torch.rand(N, 3, H, W, generator=g) * 2 - 1}
    # ---- oracle composition of test_generator.py:144-219
    with torch.no_grad():
        lo = (256, 192)
        i1 = torch.cat([O.resize_bilinear(inp["cloth"], size=lo), O.resize_nearest(inp["cloth_mask"], lo)], 1)
        i2 = torch.cat([O.resize_nearest(inp["parse_agnostic"], lo), O.resize_bilinear(inp["densepose"], size=lo)], 1)
        flow_list, seg, wc_p, wcm_p = O.tocg_forward(sd_t, i1, i2)
        gauss, lab_w, parse_w = O.parse_glue(seg, wcm_p, H, W, "warp_grad")
        wc, wm = O.hires_warp(flow_list[-1], inp["cloth"], inp["cloth_mask"])
        wm = O.remove_overlap(F.softmax(gauss, dim=1), wm)
        wc = wc * wm + torch.ones_like(wc) * (1 - wm)
        want = O.spade_generator_forward(sd_g, torch.cat((inp["agnostic"], inp["densepose"], wc), 1), parse_w, H, W, "most")
    tocg.cuda()
    gen.cuda()
    res = tryon_step(opt, tocg, gen, {k: v.cuda() for k, v in inp.items()})
    lab_g = res["fake_parse"].cpu()[:, 0]
    frac_mism = (lab_g != lab_w).float().mean().item()
    assert frac_mism < 1e-4, f"label mismatch fraction {frac_mism}"
    assert (res["warped_cloth"].cpu() - wc).abs().max() < 1e-3
    err = (res["output"].cpu() - want).abs()
    # a flipped label changes the SPADE input locally; everything else must agree closely
    assert (err > 1e-3).float().mean().item() < 1e-3, f"{(err > 1e-3).float().mean().item()} of output pixels off by >1e-3"
    assert err.median() < 1e-5


def test_entry_scripts_on_disk_dataset(tmp_path):
    """BASELINE configs[0] plumbing on the HIP path: test_generator.py over a synthetic VITON-HD-layout data set
    on disk (256x192, 'more', batch 1, random-init tocg + generator, 4 pairs) through the torchvision-free
    pipeline, then two train_condition.py iterations on the same tree; the outputs are JPEG bytes under .png names
    (utils.py:93-109)."""
    import sys
    from PIL import Image
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd.cp_dataset import write_synthetic_dataset
    import test_generator as tg
    import train_condition as tc
    root = str(tmp_path / "data")
    write_synthetic_dataset(root, n=4, seed=1)
    write_synthetic_dataset(root, n=4, datamode="train", list_name="train_pairs.txt", seed=2)
    out = str(tmp_path / "out")
    torch.manual_seed(0)
    tg.main(["--dataroot", root, "--datamode", "test", "--data_list", "test_pairs.txt", "--fine_height", "256",
             "--fine_width", "192", "--num_upsampling_layers", "more", "-b", "1", "-j", "0", "--random_init_tocg",
             "--gen_checkpoint", "", "--tocg_ngf", "16", "--ngf", "8", "--output_dir", out, "--datasetting", "unpaired"])
    files = sorted(f for f in __import__("os").listdir(out))
    assert len(files) == 4 and all(f.endswith(".png") for f in files), files
    im = Image.open(__import__("os").path.join(out, files[0]))
    assert im.format == "JPEG" and im.size == (192, 256)
    ck = str(tmp_path / "ck")
    tc.main(["--dataroot", root, "--datamode", "train", "--data_list", "train_pairs.txt", "-b", "2", "-j", "0",
             "--Ddownx2", "--lasttvonly", "--interflowloss", "--max_steps", "2", "--display_count", "1", "--ngf", "8",
             "--checkpoint_dir", ck, "--name", "t", "--shuffle", "--vgg_random_init"])
    sd = torch.load(__import__("os").path.join(ck, "t", "tocg_final.pth"), map_location="cpu")
    assert "ClothEncoder.0.block.1.running_mean" in sd and int(sd["out_layer.block.1.num_batches_tracked"]) == 2


def test_gt_branch_of_generator_inputs():
    """--GT (train_generator.py:253-274): parse7 = merge(one-hot(argmax(parse_GT))), x = cat(agnostic, pose, parse_cloth)."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import ops, pipeline
    g = torch.Generator().manual_seed(2)
    N, H, W = 2, 64, 48
    lab = torch.randint(0, 13, (N, 1, H, W), generator=g)
    inputs = {"parse": torch.zeros(N, 13, H, W).scatter_(1, lab, 1.0).cuda(),
              "agnostic": torch.rand(N, 3, H, W, generator=g).cuda(), "densepose": torch.rand(N, 3, H, W, generator=g).cuda(),
              "parse_cloth": torch.rand(N, 3, H, W, generator=g).cuda()}
    x, parse7 = pipeline.make_generator_inputs(Namespace(GT=True), None, inputs)
    want = torch.zeros(N, 7, H, W)
    for i, src in O.PARSE_MERGE.items():
        for l in src:
            want[:, i] += (lab[:, 0] == l).float()
    assert torch.equal(ops.to_nchw(parse7).cpu(), want)
    assert torch.equal(x.cpu(), torch.cat([inputs["agnostic"], inputs["densepose"], inputs["parse_cloth"]], 1).cpu())


def test_hipgraph_replay_is_bit_identical_to_eager():
    """hr_viton_amd.graph: the captured try-on step (tocg -> glue -> generator) replays the same launches, so the
    outputs are bit-identical to an eager call on the same inputs -- also after the inputs change (static-buffer
    refresh), and the integer label map is exact."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd.graph import graphed_condition, graphed_tryon
    from hr_viton_amd.network_generator import SPADEGenerator
    from hr_viton_amd.networks import ConditionGenerator
    from hr_viton_amd.ops import HrvError
    from hr_viton_amd.pipeline import tryon_step
    H, W = 256, 192
    opt = Namespace(cuda=True, warp_feature="T1", out_layer="relu", norm_G="spectralaliasinstance", gen_semantic_nc=7,
                    ngf=8, num_upsampling_layers="more", fine_height=H, fine_width=W, occlusion=True,
                    clothmask_composition="warp_grad")
    torch.manual_seed(0)
    tocg = ConditionGenerator(opt, 4, 16, 13, ngf=16, norm_layer=nn.BatchNorm2d).cuda().eval()
    gen = SPADEGenerator(opt, 9)
    gen.init_weights("xavier", 0.02)
    gen.cuda().eval()          # noise_scale is zero at init: the SPADE noise draw does not reach the output

    def batch(seed, N=2):
        g = torch.Generator().manual_seed(seed)
        lab = torch.randint(0, 13, (N, 1, H // 32, W // 32), generator=g).repeat_interleave(32, 2).repeat_interleave(32, 3)
        b = {"cloth": torch.rand(N, 3, H, W, generator=g) * 2 - 1, "cloth_mask": (torch.rand(N, 1, H, W, generator=g) > 0.4).float(),
             "parse_agnostic": torch.zeros(N, 13, H, W).scatter_(1, lab, 1.0), "densepose": torch.rand(N, 3, H, W, generator=g) * 2 - 1,
             "agnostic": torch.rand(N, 3, H, W, generator=g) * 2 - 1}
        return {k: v.cuda() for k, v in b.items()}

    b0, b1 = batch(0), batch(1)
    g = graphed_tryon(opt, tocg, gen, b0)
    for b in (b0, b1, b0):
        want = tryon_step(opt, tocg, gen, b)
        got = g(b)
        for k in ("output", "warped_cloth", "warped_clothmask", "fake_segmap"):
            assert torch.equal(got[k], want[k]), k
        assert torch.equal(got["fake_parse"], want["fake_parse"])
        for fg, fw in zip(got["flow_list"], want["flow_list"]):
            assert torch.equal(fg, fw)
    assert g.replays == 3
    with pytest.raises(HrvError):
        g({k: v[:1] for k, v in b0.items()})          # a new shape needs a new capture
    with pytest.raises(HrvError):
        graphed_condition(opt, tocg, torch.zeros(1, 4, H, W), torch.zeros(1, 16, H, W))   # host tensors: no CPU path
    # the condition generator alone
    i1, i2 = torch.randn(2, 4, H, W, device="cuda"), torch.randn(2, 16, H, W, device="cuda")
    gc = graphed_condition(opt, tocg, i1, i2)
    want = tocg(opt, i1, i2)
    got = gc({"input1": i1, "input2": i2})
    assert torch.equal(got[1], want[1]) and torch.equal(got[2], want[2]) and torch.equal(got[0][-1], want[0][-1])
    # a weight change rebuilds the module's inference plan (new packed streams): the graph still reads the old plan's buffers --
    # kept alive by the guard, but stale -- so a replay is refused, loudly (it used to read freed memory silently)
    with torch.no_grad():
        tocg.out_layer.block[4].bias.add_(0.5)
    with pytest.raises(HrvError, match="rebuilt its inference plan"):      # (changed, no eager call yet: the parameter versions moved)
        gc({"input1": i1, "input2": i2})
    tocg(opt, i1, i2)                                  # eager call: the plan is rebuilt for the new weights
    with pytest.raises(HrvError, match="rebuilt its inference plan"):
        gc({"input1": i1, "input2": i2})
    with pytest.raises(HrvError, match="rebuilt its inference plan"):
        g(b0)
