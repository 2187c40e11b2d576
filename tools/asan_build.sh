#!/bin/bash
# Host-side AddressSanitizer + LeakSanitizer build of the library and of tests/cpp/abi_stress.cpp
# into build_asan/ (device code is compiled as usual: GPU ASan is not available on this pool).
# Run build_asan/abi_stress_asan on a GPU box; it must print "abi stress passed" and no ASan report.
set -e
cd "$(dirname "$0")/.."
mkdir -p build_asan && cd build_asan
H=/opt/rocm/bin/hipcc
F="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer"
for f in ppo_xp device_shuffle rnorm; do $H $F -c -o $f.o ../ppo_amd/csrc/$f.hip -I../ppo_amd/csrc; done
$H --offload-arch=gfx950 -shared -fsanitize=address -fno-gpu-sanitize -o libppo_xp.so ppo_xp.o device_shuffle.o rnorm.o -ldl
rm -f *.o
C=/opt/rocm/lib/llvm/bin/clang++
for t in abi_stress test_reference_cases; do
  $C -O1 -g -std=c++17 -fsanitize=address -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I../include -I/opt/rocm/include \
     -o ${t}_asan ../tests/cpp/$t.cpp -L. -lppo_xp -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib
done
ls -la
