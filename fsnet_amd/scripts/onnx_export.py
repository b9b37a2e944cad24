"""ONNX export driver (mirror of the reference's scripts/onnx_export.py:13-69): config -> meta-arch -> checkpoint
(strict=False) -> eval() -> forward := dummy_forward -> torch.onnx.export(opset 11) -> read the file back, check it,
print it, evaluate it once.

    python -m fsnet_amd.scripts.onnx_export --config CFG --checkpoint_path CKPT [--onnx_file metaarch.onnx]
                                            [--input_names input] [--output_names output] [--gpu 0]

The reference loads the file with `onnx`, checks it with `onnx.checker`, and runs it under onnxruntime; neither
package is in the ROCm image, so the read-back / structural check / printout come from fsnet_amd/export/onnx_graph.py
and the run is the HIP engine's own `dummy_forward` on the same dummy input (the shapes the file declares are
compared with what the engine returns).  With `onnx` / `onnxruntime` installed the reference's steps run as well."""
import argparse

import torch

from fsnet_amd.export import onnx_graph
from fsnet_amd.vision_base.networks.utils.utils import load_models
from fsnet_amd.vision_base.utils.builder import build
from fsnet_amd.vision_base.utils.utils import cfg_from_file


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="config/config.py")
    ap.add_argument("--checkpoint_path", default="monodepth.pth")
    ap.add_argument("--onnx_file", default="metaarch.onnx")
    ap.add_argument("--input_names", nargs="+", default=["input"])
    ap.add_argument("--output_names", nargs="+", default=["output"])
    ap.add_argument("--gpu", type=int, default=0)
    args = ap.parse_args(argv)
    cfg = cfg_from_file(args.config)
    cfg.trainer.gpu = args.gpu
    torch.cuda.set_device(cfg.trainer.gpu)
    meta_arch = build(**cfg.meta_arch).cuda()
    load_models(args.checkpoint_path, meta_arch, map_location="cuda:%d" % args.gpu, strict=False)
    meta_arch.eval()
    print("Loaded model from %s." % args.checkpoint_path)

    dummy_input = torch.zeros([1, cfg.data.rgb_shape[2], cfg.data.rgb_shape[0], cfg.data.rgb_shape[1]]).cuda()
    onnx_graph.export(meta_arch, dummy_input, args.onnx_file, args.input_names, args.output_names, opset_version=11)
    print("Finish export, start checking the exported file %s." % args.onnx_file)

    model = onnx_graph.read_model(args.onnx_file)
    onnx_graph.check_model(model)
    print("Finish onnx checker check.")
    print("-----------------onnx helper print-----------------")
    print(onnx_graph.printable_graph(model))
    print("Finish onnx helper print.")
    try:
        import onnx
        onnx.checker.check_model(onnx.load(args.onnx_file))
        print("onnx.checker agrees.")
    except ImportError:
        pass

    with torch.no_grad():
        outputs = meta_arch.dummy_forward(dummy_input)
    depth = outputs["depth"]
    declared = model["graph"]["outputs"][0]["shape"]
    if len(declared) != depth.dim() or any(isinstance(d, int) and d != s for d, s in zip(declared, depth.shape)):
        raise RuntimeError("exported output shape %s vs engine output %s" % (declared, tuple(depth.shape)))
    try:
        import onnxruntime as ort
        sess = ort.InferenceSession(args.onnx_file, providers=["ROCMExecutionProvider", "CPUExecutionProvider"])
        got = sess.run(None, {args.input_names[0]: dummy_input.cpu().numpy()})[0]
        print("onnxruntime vs HIP engine: max abs deviation %.3e" % float(abs(got - depth.cpu().numpy()).max()))
    except ImportError:
        pass
    print("The actual output of the HIP engine: outputs[0].shape=%s" % (tuple(depth.shape),))
    return model


if __name__ == "__main__":
    main()
