import sys; sys.path.insert(0, '.')
import torch
from oracle import fsnet_oracle as O
from tests.test_model_gpu import build_model, to_dev
dev = torch.device('cuda:0')
for (B,H,W) in ((2,64,128),(4,96,320)):
    sd0 = O.init_state(seed=3, with_pose=True)
    data = O.synthetic_batch(B, H, W, seed=7)
    sd = {k: v.clone() for k, v in sd0.items()}
    fo = O.resnet_forward(sd, "depth_backbone.", data[("image", 0)])
    oo = O.depth_decoder_forward(sd, "head.depth_decoder.", fo, 0.5, 100.0)
    for dt in (torch.float32, torch.bfloat16):
        m = build_model(True, H, W, dev, dt, sd0)
        feats = m.depth_backbone(data[("image", 0)].to(dev))
        outs = m.head.forward_depth(feats)
        for s in range(4):
            ref = oo[("disp", s)].detach()
            r = ((outs[("disp", s)].detach().cpu() - ref).abs() / ref.abs())
            print(B,H,W,dt, "scale", s, "disp rel max %.4f mean %.5f" % (float(r.max()), float(r.mean())))
        f4 = feats[4].float().cpu(); r4 = fo[4].detach()
        print("  feat4 rel-L2", float((f4-r4).norm()/r4.norm()))
        m2 = build_model(True, H, W, dev, dt, sd0)
        out = m2(to_dev(data, dev), dict(is_training=True)); out["loss"].backward(); torch.cuda.synchronize()
        tr = O.OracleTrainer(sd0, with_pose=True, clip=None)
        total, ld, _, raw, _ = tr.step(data)
        print("  loss", float(out["loss"]), float(total), "rel", abs(float(out["loss"])-float(total))/float(total))
        rels = []
        for k, p in m2.named_parameters():
            ref = raw[k]
            if ref.norm() < 1e-3 * max(r.norm() for r in raw.values()): continue
            rels.append((float((p.grad.cpu()-ref).norm()/ref.norm()), k))
        rels.sort(reverse=True)
        print("  worst grad rel-L2:", rels[:3], "median", rels[len(rels)//2][0])
