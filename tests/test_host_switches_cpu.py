"""Host-side selection logic added in round 5 that needs no GPU: which PatchGAN convolutions keep fp32 operands (gen_train._d_f32), the
weight-gradient side stream's switch (train_ops.wgrad_side_maxpix / the no-op joins of a process that never forked), eviction of a
replaced serving plan's packed streams (train_ops.evict_serving_packs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _mods():
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import gen_train, train_ops
    return gen_train, train_ops


def test_patchgan_fp32_layer_selection(monkeypatch):
    G, _ = _mods()
    for k in ("HRV_D_F32_MASK", "HRV_D_F32_LAYERS", "HRV_D_F32_PARTS", "HRV_D_F32_SCOPE", "HRV_D_F32_SCALES"):
        monkeypatch.delenv(k, raising=False)
    # default (round 6): the FORWARDS of model0 .. model2 of discriminator_1 (the half-resolution scale), in the discriminator's own step only
    assert G._d_f32(1, "fwd", True, 1) and not G._d_f32(1, "bwd", True, 1) and not G._d_f32(1, "fwd", False, 1)
    assert not G._d_f32(1, "fwd", True, 0)
    assert [G._d_f32(i, "fwd", True, 1) for i in range(4)] == [True, True, True, False]
    monkeypatch.setenv("HRV_D_F32_SCALES", "3")          # (the rest of this test: both scales)
    monkeypatch.setenv("HRV_D_F32_MASK", "0")
    assert not any(G._d_f32(i, p, True) for i in range(4) for p in ("fwd", "bwd"))      # amp O1's choice: everything in bf16
    monkeypatch.setenv("HRV_D_F32_MASK", "6")
    monkeypatch.setenv("HRV_D_F32_PARTS", "all")
    monkeypatch.setenv("HRV_D_F32_SCOPE", "always")
    assert [G._d_f32(i, "bwd", False) for i in range(4)] == [False, True, True, False]
    monkeypatch.setenv("HRV_D_F32_MASK", "0")
    monkeypatch.setenv("HRV_D_F32_LAYERS", "2")                                         # the first two layers
    assert [G._d_f32(i, "fwd", True) for i in range(4)] == [True, True, False, False]
    monkeypatch.setenv("HRV_D_F32_PARTS", "bwd")
    assert not G._d_f32(0, "fwd", True) and G._d_f32(0, "bwd", True)


def test_engine_mode_context_restores_the_flag():
    G, T = _mods()
    old = T.MMA_BF16[0]
    try:
        T.MMA_BF16[0] = True
        with G._EngineMode(True):
            assert T.MMA_BF16[0] is False
        assert T.MMA_BF16[0] is True
        with G._EngineMode(False):
            assert T.MMA_BF16[0] is True
        try:
            with G._EngineMode(True):
                raise RuntimeError("x")
        except RuntimeError:
            pass
        assert T.MMA_BF16[0] is True
    finally:
        T.MMA_BF16[0] = old


def test_side_stream_switch_and_joins_without_a_gpu(monkeypatch):
    _, T = _mods()
    monkeypatch.delenv("HRV_WGRAD_SIDE", raising=False)
    monkeypatch.delenv("HRV_WGRAD_SIDE_MAXPIX", raising=False)
    assert T.wgrad_side_maxpix() == 0                      # opt-in: measured no gain (profiles/r05_ab_wgrad_side.txt)
    monkeypatch.setenv("HRV_WGRAD_SIDE", "1")
    assert T.wgrad_side_maxpix() == 65536
    monkeypatch.setenv("HRV_WGRAD_SIDE_MAXPIX", "200000")
    assert T.wgrad_side_maxpix() == 200000
    monkeypatch.setenv("HRV_WGRAD_SIDE", "0")
    assert T.wgrad_side_maxpix() == 0
    # a process that never forked (every CPU process): joins are no-ops and never touch torch.cuda
    T.wgrad_join()
    T.wgrad_sync_for_collective()
    with T.wgrad_side(4096):                               # switched off: the body runs inline
        pass


def test_evict_serving_packs_drops_only_the_named_plans():
    _, T = _mods()
    saved = dict(T._FROZEN_PACKS)
    try:
        T._FROZEN_PACKS.clear()
        T._FROZEN_PACKS[("p2", 1, ("serve", 5, 0), 0)] = (None, ())
        T._FROZEN_PACKS[("dev", ("serve", 5, 11), 7)] = (None, ())
        T._FROZEN_PACKS[("dev", ("serve", 6, 1), 7)] = (None, ())
        T._FROZEN_PACKS[("p2", 9, "vgg", 0)] = (None, ())
        assert T.evict_serving_packs([5]) == 2
        assert sorted(map(str, T._FROZEN_PACKS)) == ["('dev', ('serve', 6, 1), 7)", "('p2', 9, 'vgg', 0)"]
        assert T.evict_serving_packs([5]) == 0
    finally:
        T._FROZEN_PACKS.clear()
        T._FROZEN_PACKS.update(saved)
