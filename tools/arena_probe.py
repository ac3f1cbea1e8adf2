"""tools/arena_probe.py - HBM the fp32 engines hold with / without the activation arena (run on the GPU box)."""
import os, sys, subprocess
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    from hyperpose_amd import _lib
    from hyperpose_amd.engine import Engine, Model
    _lib.init(0)
    arch, w, h, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    m = Model(arch, w, h)
    e = Engine.from_model(m, m.init_weights(1), max_batch=n, dtype="f32")
    db, ai = e.device_bytes, e.arena_info
    print(f"{arch} {w}x{h} batch {n} HP_NO_ARENA={os.environ.get('HP_NO_ARENA', '0')}: activations {db['activations'] / 2**20:.1f} MiB, weights {db['weights'] / 2**20:.1f}, outputs {db['outputs'] / 2**20:.1f}; "
          f"arena: {ai['buffers']} buffers for {ai['tensors']} tensors, {ai['bytes_without_reuse'] / 2**20:.1f} MiB without re-use")
else:
    for cfg in (("lw_openpose_vggtiny", 432, 368, 1), ("lw_openpose_mobilenet", 432, 368, 8), ("openpose_vgg19", 768, 432, 16), ("pose_proposal_resnet50", 384, 384, 32), ("pifpaf_resnet50", 385, 385, 64)):
        for na in ("0", "1"):
            env = dict(os.environ, HP_NO_ARENA=na) if na == "1" else {k: v for k, v in os.environ.items() if k != "HP_NO_ARENA"}
            subprocess.call([sys.executable, __file__] + [str(x) for x in cfg], env=env)
