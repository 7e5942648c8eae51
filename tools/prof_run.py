"""One frame of a scene at a given spp, nothing else: the target command for rocprofv3 runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_amd as rt
name = sys.argv[1] if len(sys.argv) > 1 else "cu_cover2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 100
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
world, kw = rt.get_scene(name, spp=spp)
dev = rt.DeviceScene(world)
img, st = dev.render(kw["camera"], kw["pattern"], kw["background"], kw["width"], kw["height"], flags=flags)
paths = kw["width"] * kw["height"] * spp
print("%s %d spp: %.4f s, %.2f Mpaths/s" % (name, spp, st.seconds, paths / st.seconds / 1e6),
      "rays/path %.3f nodes/ray %.2f prims/ray %.2f | lane utilisation: node steps %.3f leaf tests %.3f retire/fetch %.3f" % (
          st.rays / paths, st.node_visits / max(1, st.rays), st.prim_tests / max(1, st.rays),
          st.node_visits / max(1, 64 * st.node_wave_steps), st.prim_tests / max(1, 64 * st.prim_wave_steps),
          st.rays / max(1, 64 * st.bounce_wave_steps)) if flags & 1 else "",
      "| extend %.4f s in %d launches, shade %.4f s in %d launches" % (
          st.extend_seconds, st.extend_launches, st.shade_seconds, st.shade_launches) if flags & 2 else "")
