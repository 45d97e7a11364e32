# development tool: rebuild libnsx.so with each cache-policy mask (-DNSX_NT, nsx_internal.hpp) and run a short bench; on the GPU box
set -e
mkdir -p gpurun_out
for NT in ${NT_LIST:-0 1 4 5}; do
  make -B device HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-parameter -DNSX_NT=$NT" > gpurun_out/nt_build_$NT.log 2>&1
  for L in ${LINKS_LIST:-2}; do
    NSX_MGS_LINKS=$L python bench.py --steps 6 --warmup 2 --spinup 5 --no-cpu --profile-steps 4 > gpurun_out/nt_${NT}_l$L.json 2> gpurun_out/nt_${NT}_l$L.err
    echo "NT=$NT links=$L done"
  done
done
