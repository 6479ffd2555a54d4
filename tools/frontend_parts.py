"""S2 stage entry of bench.py alone, for several sub-batch counts: python tools/frontend_parts.py [parts ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multiprocessing as mp


def render(_):
    from object_slam_amd import scene
    return scene.make_rgbd_sequence(0, 37, speed=2.0, with_masks=False)


if __name__ == "__main__":
    with mp.get_context("fork").Pool(1) as pool:      # rendered before this process touches the GPU
        q = pool.map(render, [0])[0]
    import bench
    for parts in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
        r = bench.frontend_stage(q["gray"], q["Twc"], q["depth"], 0, 20, parts=parts)
        print(json.dumps({k: r[k] for k in ("parts", "frames_per_s", "ms_per_step", "per_frame_us")}), flush=True)
