"""CPU: BASELINE configs[0] plumbing - a 512x384 image pair through the torch-CPU fp32 restatement of the
two-view network (reduced depth, seeded random weights): output contract of `reconstruct`
(mast3r_utils.py:284-294) and basic invariants.  No GPU, no reference code at run time."""
import numpy as np
import torch

from mast3r_slam import model as M, synthetic
from oracle import model as OM


def test_two_view_contract_512x384_cpu():
    cfg = dict(M.FULL_CFG, enc_depth=2, dec_depth=2, hooks=(0, 1, 2, 2))
    w = M.init_random_weights(cfg, seed=3)
    h, wd = 384, 512
    im1 = torch.from_numpy(synthetic.textured_image(h, wd, 0)[None])
    im2 = torch.from_numpy(synthetic.textured_image(h, wd, 1)[None])
    torch.set_num_threads(8)
    with torch.no_grad():
        o1, o2 = OM.reconstruct(w, im1, im2, cfg)
    for o in (o1, o2):
        assert o["pts3d"].shape == (1, h, wd, 3) and o["conf"].shape == (1, h, wd)
        assert o["desc"].shape == (1, h, wd, 24) and o["desc_conf"].shape == (1, h, wd)
        assert all(torch.isfinite(v).all() for v in o.values())
        assert float((o["desc"].norm(dim=-1) - 1).abs().max()) < 1e-5      # unit descriptors
        assert float(o["conf"].min()) > 1.0 and float(o["desc_conf"].min()) > 0.0
    assert not torch.equal(o1["pts3d"], o2["pts3d"])                        # two heads, two views
    # self-pair symmetry of the encoder: identical images give identical encoder tokens
    f, pos = OM.encode(w, torch.cat([im1, im1]), cfg)
    assert torch.equal(f[0], f[1]) and pos.shape == (24 * 32, 2)


def test_weight_table_and_flop_count():
    w = M.init_random_weights(M.TINY_CFG, seed=0)
    # matrices are bf16-representable (the product stores them in bf16)
    k = "enc_blocks.0.attn.qkv.weight"
    assert torch.equal(w[k], w[k].to(torch.bfloat16).float()) and w[k].shape == (3072, 1024)
    assert w["downstream_head1.head_local_features.fc2.weight"].shape == (25 * 256, 4 * 1792)
    # algorithmic FLOPs of the full model at 512x512 (SURVEY 8d estimate: ~2.81 TFLOP/pair)
    class _Shim:                                                            # flops_per_pair only needs cfg
        cfg = M.FULL_CFG
    fl = M.Mast3rFull.flops_per_pair(_Shim(), 512, 512)
    assert 2.7e12 < fl < 2.9e12
