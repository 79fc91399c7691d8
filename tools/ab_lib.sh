#!/bin/bash
# tools/ab_lib.sh <name> <source.hip> <extra hipcc flags…>: an A/B build of libgmsx.so in gms_amd/lib_<name>/ that differs from gms_amd/lib/ in ONE
# kernel file compiled with extra flags and -DGMSX_DEV_HOOKS — the only way to the "wrong counts" A/B switches and to GMSX_TC_ONLY: `make` never sets it —
# (the other objects are reused); select it with GMSX_LIB=gms_amd/lib_<name>/libgmsx.so
set -e
cd "$(dirname "$0")/../gms_amd/csrc"
name=$1; src=$2; shift 2
out=../lib_$name
mkdir -p $out/obj
base=$(basename $src .hip)
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -I../../include -Ihost -Ihip -DGMSX_DEV_HOOKS "$@" -c hip/$base.hip -o $out/obj/$base.o
objs=$(ls ../lib/obj/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libgmsx.so $objs $out/obj/$base.o -fopenmp -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
echo built $out/libgmsx.so
