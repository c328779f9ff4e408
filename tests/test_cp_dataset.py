"""CPU: the torchvision-free VITON-HD data pipeline (hr_viton_amd.cp_dataset) against one sample produced by
the REAL reference CPDatasetTest on the same synthetic on-disk data set (golden: oracle/make_golden.py dataset),
plus the structural invariants the training scripts rely on."""
from argparse import Namespace

import torch

from conftest import load_golden


def _opt(root, h=64, w=48, **kw):
    d = dict(dataroot=root, datamode="test", data_list="test_pairs.txt", fine_height=h, fine_width=w, semantic_nc=13,
             shuffle=False, batch_size=2, workers=0)
    d.update(kw)
    return Namespace(**d)


def test_item_matches_reference_dataset(tmp_path):
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import cp_dataset as P
    g = load_golden("cpdataset_item1_64x48.pt")["item"]
    P.write_synthetic_dataset(str(tmp_path), n=2, seed=0)
    item = P.CPDatasetTest(_opt(str(tmp_path)))[1]
    assert set(item.keys()) == set(g.keys())
    assert item["im_name"] == g["im_name"] and item["c_name"] == g["c_name"]
    for k, want in g.items():
        if isinstance(want, dict):
            for kk, ww in want.items():
                if torch.is_tensor(ww):
                    assert item[k][kk].dtype == ww.dtype and item[k][kk].shape == ww.shape, (k, kk)
                    assert torch.equal(item[k][kk], ww), (k, kk, (item[k][kk] - ww).abs().max())
        elif torch.is_tensor(want):
            assert item[k].dtype == want.dtype and item[k].shape == want.shape, k
            assert torch.equal(item[k], want), (k, (item[k] - want).abs().max())
    # the agnostic image really differs from the person image (torso painted gray) but keeps the head
    assert (item["agnostic"] - item["image"]).abs().max() > 0.2


def test_invariants_and_loader(tmp_path):
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import cp_dataset as P
    P.write_synthetic_dataset(str(tmp_path), n=4, datamode="train", list_name="train_pairs.txt", seed=3)
    opt = _opt(str(tmp_path), 128, 96, datamode="train", data_list="train_pairs.txt")
    ds = P.CPDataset(opt)
    assert len(ds) == 4
    it = ds[0]
    assert it["im_name"].startswith("image/") and set(it["cloth"].keys()) == {"paired"}
    assert it["cloth"]["paired"].shape == (3, 128, 96) and it["cloth_mask"]["paired"].shape == (1, 128, 96)
    assert set(it["cloth_mask"]["paired"].unique().tolist()) <= {0.0, 1.0}
    assert it["parse"].shape == (13, 128, 96) and torch.all(it["parse"].sum(0) == 1)
    assert torch.equal(it["parse"].argmax(0, keepdim=True).float(), it["parse_onehot"])
    assert torch.equal(it["pcm"], it["parse"][3:4]) and it["pcm"].sum() > 0
    assert torch.all(it["parse_agnostic"].sum(0) == 1) and it["parse_agnostic"][3].sum() == 0   # cloth removed
    for k in ("densepose", "pose", "image", "agnostic", "parse_cloth"):
        assert it[k].shape == (3, 128, 96) and it[k].min() >= -1 and it[k].max() <= 1, k
    loader = P.CPDataLoader(opt, ds)
    b1 = loader.next_batch()
    loader.next_batch()
    b3 = loader.next_batch()     # wraps around after 2 batches of 2
    assert b1["cloth"]["paired"].shape == (2, 3, 128, 96) and b3["parse"].shape == (2, 13, 128, 96)
    assert len(b1["c_name"]["paired"]) == 2


def test_data_parallel_loaders_draw_disjoint_shards(tmp_path):
    """One process per GPU: the ranks' loaders cut ONE shared per-epoch permutation into disjoint shards
    (DistributedSampler), reshuffled on wrap-around -- no sample twice within an epoch across the ranks."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import cp_dataset as P
    P.write_synthetic_dataset(str(tmp_path), n=8, datamode="train", list_name="train_pairs.txt", seed=5)
    opt = _opt(str(tmp_path), 64, 48, datamode="train", data_list="train_pairs.txt")
    ds = P.CPDataset(opt)
    loaders = [P.CPDataLoader(opt, ds, rank=r, world=2) for r in range(2)]
    epochs = []
    for _ in range(2):                      # 8 samples / 2 ranks / batch 2 = 2 batches per rank and epoch
        names = [[n for _ in range(2) for n in ld.next_batch()["im_name"]] for ld in loaders]
        assert not set(names[0]) & set(names[1])
        assert len(set(names[0]) | set(names[1])) == 8
        epochs.append(names)
    assert epochs[0] != epochs[1]           # set_epoch: a new permutation after the wrap-around
