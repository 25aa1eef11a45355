import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle import nets, resnet
from gpu_util import rel_l2, randomize_bn, cpu_state
from agplace_amd.models_baseline.dbvanilla2d import DBVanilla2D
from agplace_amd.options import Options
from agplace_amd import ops
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
for prec in (2, 3, 4):
    for hw in (64, 224):
        opt = Options(dbimage_fe="resnet50", dbimage_fe_layers="3_4_6", mfma_precision=prec)
        torch.manual_seed(51)
        md = randomize_bn(DBVanilla2D("db", 256, opt=opt), seed=4).to(dev).eval()
        tiles = torch.randn(2, 1, 3, hw, hw, generator=torch.Generator().manual_seed(53))
        pd = {k: (v.double() if v.is_floating_point() else v) for k, v in cpu_state(md).items()}
        fe = md.dbimage_fes[0]
        maps = fe.forward_maps(tiles[:, 0].to(dev), prec=prec)
        pfe = {k[len("dbimage_fes.0.fe."):]: v for k, v in pd.items() if k.startswith("dbimage_fes.0.fe.")}
        ref = resnet.forward_resnet(tiles[:, 0].double(), pfe, "resnet50", 3)
        errs = [rel_l2(m.to_f32(), r) for m, r in zip(maps, ref)]
        v = md.dbimage_pools[0].pool_map(maps[-1])
        rv = nets.gem(ref[-1], pd["dbimage_pools.0.p"]).flatten(1)
        e = md({"db_map": tiles.to(dev)}, mode="db")["embedding"]
        r = nets.dbvanilla2d_forward_db({"db_map": tiles.double()}, pd, opt)["embedding"]
        mlp = md.dbimage_mlps[0](v)
        rmlp = nets.db_mlp(rv, pd, "dbimage_mlps.0.")
        print(f"prec {prec} hw {hw}: maps {['%.1e' % x for x in errs]} gem {rel_l2(v, rv):.1e} mlp {rel_l2(mlp, rmlp):.1e} emb {rel_l2(e, r):.1e}")
