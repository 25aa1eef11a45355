"""Host-side logic that needs no GPU: options, parameter names, time grid, loud failures."""
import types

import numpy as np
import pytest
import torch

import agplace_amd
from agplace_amd import ops, parallel
from agplace_amd.options import Options, from_reference_opt
from oracle import nets, ode, resnet


def test_options_defaults_match_reference_flags():
    o = Options()
    assert (o.mm_imgfe, o.mm_imgfe_layers, o.mm_stg2fuse_dim) == ("resnet18", "2_2_2", 256)
    assert (o.diff_type, o.diff_direction, o.odeint_method, o.odeint_size) == ("fcode@relu", "backward", "euler", 0.1)
    assert o.output_type == ["image", "vox", "shallow"] and o.final_fusetype == "add" and o.final_l2 is False
    assert (o.imagevoxorg_weight, o.shalloworg_weight, o.stg2imagevox_weight, o.stg2fuse_weight) == (0.0, 1.0, 0.1, 0.0)
    assert o.recall_values == [1, 5, 10, 20] and o.features_dim == 256
    ns = types.SimpleNamespace(odeint_method="rk4", odeint_size=0.25, final_type="imageorg_stg2image",
                               output_type="image_vox_shallow", unrelated_flag=1)
    a = from_reference_opt(ns)
    assert a.odeint_method == "rk4" and a.final_type == ["imageorg", "stg2image"]


def test_time_grid_equals_oracle_grid():
    for step in (0.1, 0.25, 0.3, 0.5, 1.0, 0.07):
        assert ops.ode_grid_dts(step) == ode.grid_dts(step).tolist()


def test_state_dict_keys_match_reference_names():
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.models_baseline.dbvanilla2d import DBVanilla2D
    o = Options()
    mm_keys = set(MM(opt=o).state_dict().keys())
    assert mm_keys == set(nets.init_mm_params(o).keys())
    db_keys = set(DBVanilla2D("db", 256, opt=o).state_dict().keys())
    assert db_keys == set(nets.init_db_params(o).keys())
    # spot-check the names train.py / checkpoints rely on
    for k in ("image_fe.fe.conv1.weight", "image_fe.fe.layer2.0.downsample.0.weight", "image_fe.fe.fc.weight",
              "image_pool.p", "fuseblocktoshallow.blocks.2.blocks.0.func.func.fc.weight",
              "fuseblocktoshallow.updimsvox.1.bias", "stg2fuseblock.projsfuseimg.0.0.weight",
              "stg2fuseblock.projsimgfuse.0.0.weight", "stg2fuseblock.ffnsimg.0.conv1.bias",
              "stg2fuseblock.ffnsfuse.0.ffns.0.ln2.weight", "stg2fuseblock.poolimage.p", "stg2fusefc.bias",
              "stg2image_weight"):
        assert k in mm_keys, k
    assert "image_fe.fe.layer4.0.conv1.weight" not in mm_keys      # layer4 -> Identity (image_fe.py:26)


def test_optimizer_group_attributes_exist():
    from agplace_amd.network_mm.mm import MM
    m = MM(opt=Options())
    for a in ("image_fe", "image_pool", "fuseblocktoshallow", "stg2fuseblock", "stg2fusefc", "image_weight",
              "vox_weight", "shallow_weight", "imageorg_weight", "voxorg_weight", "shalloworg_weight",
              "stg2image_weight", "stg2vox_weight", "stg2fuse_weight"):
        assert hasattr(m, a), a
    assert not m.stg2image_weight.requires_grad     # learnweight flags default False (options.py:139-146)


def test_reference_optimizer_layout():
    """train.py:165-190, 213-214: one Adam over the database model at lrdb, one over the query model's 16 groups at lr / lrpc."""
    from agplace_amd.models_baseline.dbvanilla2d import DBVanilla2D
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.train_fns import reference_optimizers
    mq, mdb = MM(opt=Options()), DBVanilla2D("db", 256, opt=Options())
    odb, oq = reference_optimizers(mdb, mq)
    assert len(odb.param_groups) == 1 and odb.param_groups[0]["lr"] == 1e-5
    assert len(oq.param_groups) == 16
    assert [g["lr"] for g in oq.param_groups] == [1e-5, 1e-5, 1e-4, 1e-4, 1e-5, 1e-5, 1e-5, 1e-5, 1e-4] + [1e-5] * 7
    ids_q = [id(p) for g in oq.param_groups for p in g["params"]]
    assert sorted(ids_q) == sorted(id(p) for p in mq.parameters())           # every query parameter, once
    assert sorted(id(p) for p in odb.param_groups[0]["params"]) == sorted(id(p) for p in mdb.parameters())


def test_reference_error_behaviour():
    from agplace_amd.network_mm.ffns import FC, select_act
    from agplace_amd.network_mm.diff_block import DiffBlock
    from agplace_amd.network_mm.image_fe import ImageFE
    with pytest.raises(NotImplementedError):
        select_act("gelu")
    with pytest.raises(NotImplementedError):
        FC(8, 8, "gelu")
    with pytest.raises(NotImplementedError):
        DiffBlock(256, 256, opt=Options(diff_type="fcsde@relu"))
    with pytest.raises(NotImplementedError):
        ImageFE("resnet50", "2_2_2")      # the query-side ImageFE has no resnet50 branch
    from agplace_amd.network.image_fe import ImageFE as DBImageFE
    assert DBImageFE("resnet50", "3_4_6").last_dim == 1024


def test_product_path_has_no_cpu_fallback():
    from agplace_amd.network_mm.image_pooling import GeM
    with pytest.raises(RuntimeError, match="GPU"):
        GeM()(torch.rand(1, 8, 4, 4))
    with pytest.raises(RuntimeError, match="GPU"):
        ops.l2normalize(torch.rand(2, 8))


def test_shard_range_partitions():
    for n, w in ((10, 3), (8, 8), (5, 8), (100000, 8), (0, 2)):
        spans = [parallel.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def test_fold_bn_matches_batchnorm_eval():
    torch.manual_seed(0)
    c = 8
    w, b, m, v = torch.rand(c) + 0.5, torch.randn(c), torch.randn(c), torch.rand(c) + 0.5
    bias = torch.randn(c)
    s, t = ops.fold_bn(w, b, m, v, 1e-5, conv_bias=bias)
    x = torch.randn(3, c, 4, 4)
    ref = torch.nn.functional.batch_norm(x + bias.view(1, c, 1, 1), m, v, w, b, False, 0.0, 1e-5)
    np.testing.assert_allclose((x * s.view(1, c, 1, 1) + t.view(1, c, 1, 1)).numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)


def test_netvlad_init_params_against_reference_fixture(golden):
    """NetVLAD.init_params (host arithmetic; reference model/aggregation.py:112-124, fixture from the reference's own method)."""
    import numpy as np
    import torch
    from agplace_amd.model.aggregation import NetVLAD
    g = golden("netvlad_init")
    nv = NetVLAD(clusters_num=8, dim=32)
    nv.init_params(g["centroids_in"], g["descriptors"])
    assert abs(nv.alpha - float(g["alpha"])) < 1e-5 * float(g["alpha"])
    np.testing.assert_allclose(nv.conv.weight.detach().numpy(), g["conv_w"], rtol=2e-5, atol=1e-6)
    np.testing.assert_array_equal(nv.centroids.detach().numpy(), g["centroids"])
    assert nv.conv.bias is None and tuple(nv.conv.weight.shape) == (8, 32, 1, 1)
    import pytest
    with pytest.raises(ValueError):
        nv.init_params(g["centroids_in"][:4], g["descriptors"])


def test_five_crop_recall_methods_against_reference_fixture(golden):
    """retrieval.merge_crops (host arithmetic of reference test.py:35-70) on an exact brute-force search, against the recalls the
    reference's own compute_recall(test_method='nearest_crop' / 'maj_voting') produced (make_golden.py section 6)."""
    import types
    import numpy as np
    from agplace_amd import retrieval
    from oracle import knn
    g = golden("recall_crops")
    D, I, _ = knn.knn_l2_fp64(g["q5"], g["db"], 20)
    positives = [p for p in g["positives"]]

    class DS:
        queries_num = 40

        def get_positives(self):
            return positives
    for tm in ("nearest_crop", "maj_voting"):
        args = types.SimpleNamespace(features_dim=256, recall_values=[1, 5, 10, 20], majority_weight=float(g["majority_weight"]))
        pred = retrieval.merge_crops(args, D.astype(np.float32), I.copy(), DS(), tm)
        assert pred.shape == (40, 20) and all(len(set(r.tolist())) == 20 for r in pred)
        rec, _ = retrieval.recall_from_predictions(args, pred, DS())
        np.testing.assert_allclose(rec, g[tm + "_recalls"])


def test_gradbuckets_keeps_a_parameter_whose_gradient_was_written_without_a_notification():
    """a producer that writes a gradient into the flat view in place (or replaces .grad) without notifying must not
    get its parameter classed 'silent everywhere' and moved out of the exchange when the bucket order is learned; a parameter
    that really has no gradient is moved out; rebuild() honours reorder=False."""
    import torch
    from agplace_amd import parallel
    p1, p2, p3, p4 = (torch.nn.Parameter(torch.ones(3)) for _ in range(4))
    gb = parallel.GradBuckets([p1, p2, p3, p4], bucket_mb=1e-6)
    gb.zero_grad()
    (p1 * 2).sum().backward()                    # autograd hook
    p2.grad.add_(1.0)                            # in place into the view, no notification
    p3.grad = torch.full((3,), 5.0)              # view replaced, no notification
    gb.finish()
    excluded = {id(gb.params[i]) for i in gb.excluded}
    assert id(p4) in excluded and id(p2) not in excluded and id(p3) not in excluded and id(p1) not in excluded
    assert torch.equal(p3.grad, torch.full((3,), 5.0)) and p3.grad.data_ptr() == gb.flat.data_ptr() + 4 * gb.slice_of[id(p3)][0]
    # a gradient that later appears for the excluded parameter raises instead of being folded away silently
    gb.zero_grad()
    (p1 * 2).sum().backward()
    p4.grad = torch.ones(3)
    import pytest
    with pytest.raises(RuntimeError):
        gb.finish()
    gb.close()
    gb2 = parallel.GradBuckets([p1, p2], reorder=False)
    gb2.rebuild()
    assert gb2.reorder_pending is False
    gb2.close()
