#!/bin/bash
# Builds the library's sources with AddressSanitizer and UndefinedBehaviorSanitizer on the host side (device code as
# always) together with driver.cpp and runs the driver on scene files saved from the sample scenes and on the OBJ asset.
#   usage: tools/host_sanitize/run.sh [mutations per file] [seed]
set -e
here=$(cd "$(dirname "$0")" && pwd); root=$(cd "$here/../.." && pwd)
out=/tmp/rt_host_sanitize; mkdir -p $out/obj $out/data
cd "$root"
python - <<PY
import sys, numpy as np
sys.path.insert(0, "$root")
import raytracing_amd as rt
for name in ("cu_cover2", "teapot", "cornell_smoke_box", "cover1", "cover2"):
    world, kw = rt.get_scene(name, spp=1)
    rt.save_scene("$out/data/%s.rtscene" % name, world)
world, kw = rt.get_scene("cu_cover2", spp=1, F=np.float64)
rt.save_scene("$out/data/cu_cover2.rtscene64", world, F=np.float64)
PY
cp raytracing_amd/assets/teapot.obj $out/data/
SAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=undefined -Xarch_host -fno-omit-frame-pointer"
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function"
cd raytracing_amd/csrc
pids=()
for src in rt_kernels.hip rt_f64.hip accel_device.hip rt_api.cpp rt_scene.cpp bvh_build.cpp accel_build.cpp rt_output.cpp rt_obj.cpp rt_scene_file.cpp rt_perlin.cpp rt_group.cpp; do
  obj=$out/obj/${src%.*}.o
  if [ ! -f $obj ] || [ $src -nt $obj ]; then /opt/rocm/bin/hipcc $FLAGS $SAN -x hip -c $src -o $obj & pids+=($!); fi
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
/opt/rocm/bin/hipcc $FLAGS $SAN -x hip -c "$here/driver.cpp" -o $out/obj/driver.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=address -fsanitize=undefined -o $out/driver $out/obj/*.o -lz -ldl -pthread
ASAN_OPTIONS=verify_asan_link_order=0:detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 $out/driver $out/data ${1:-300} ${2:-1}
