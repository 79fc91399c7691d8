"""oracle/ref_drivers.py — TEST INFRASTRUCTURE.  Recipe that builds the REAL reference drivers (the three algorithm drivers + examples/triangle_counting.cpp) with the gmsx glue.

The driver sources are read from /root/reference where they lie, patched IN MEMORY with the `#include <gmsx_gms_glue.hpp>` line
of INTEGRATION.md §2 plus ONE driver line each, compiled from a temp file that is deleted right after, and linked against
libgmsx.so; outputs (binaries only) go to oracle/_ref/drivers/ (git-ignored, travels to the GPU box like oracle/_ref/*.so).
No reference source text is stored in this repository.

    python -m oracle.ref_drivers          # build the lean drivers (only the gmsx flavours run) into oracle/_ref/drivers/
"""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "oracle", "_ref", "drivers")  # git-ignored build output (binaries only), travels to the GPU box
LIBDIR = os.path.join(ROOT, "gms_amd", "lib")

DRIVERS = {
    "triangle_count": dict(
        src="gms/algorithms/set_based/triangle_count/triangle_count.cc",
        anchor='    benchmark_suite<RobinHoodGraph>(args, g, "RobinHoodGraph");',
        add='    benchmark_suite<HipSetGraph>(args, g, "HipSetGraph"); benchmark_suite<HipRoaringGraph>(args, g, "HipRoaringGraph");',
        # keep the run short on the GPU box: only the gmsx flavours execute
        drop=[r'\s*benchmark_suite<RoaringGraph>\(args, g, "RoaringGraph"\);', r'\s*benchmark_suite<SortedSetGraph>\(args, g, "SortedSetGraph"\);',
              r'\s*benchmark_suite<RobinHoodGraph>\(args, g, "RobinHoodGraph"\);'],
        # the triangle-count target builds the task lists inside FromCGraph = the harness's untimed "GraphExec buildTime"
        flags=["-DGMSX_ADAPTOR_UPLOAD_FLAGS=GMSX_UPLOAD_FOR_TC"]),
    # BASELINE.json configs[0]: the reference's own demonstration of plugging a new Set into SetGraph (examples/triangle_counting.cpp:62-71)
    "triangle_counting_example": dict(
        src="examples/triangle_counting.cpp",
        anchor='    BenchmarkKernelBk<BetterGraph>(args, g, TriangleCount::Seq::count_total<BetterGraph>, TriangleCount::Verify::total_count, "BetterGraph");',
        add='    BenchmarkKernelBk<HipSetGraph>(args, g, TriangleCount::Seq::count_total<HipSetGraph>, TriangleCount::Verify::total_count, "HipSetGraph");',
        # lean: SimpleSet::contains is a linear scan — minutes at scale 18; only the gmsx flavour executes on the GPU box
        drop=[r'\s*BenchmarkKernelBk<SimpleGraph>\(args, g,[^;]*;', r'\s*BenchmarkKernelBk<BetterGraph>\(args, g,[^;]*;'],
        flags=["-DGMSX_ADAPTOR_UPLOAD_FLAGS=GMSX_UPLOAD_FOR_TC"]),
    "k_clique_count": dict(
        src="gms/algorithms/set_based/k_clique_count/k_clique_count_set_based.cc",
        anchor="    return 0;",
        add=('    BenchmarkKernel(args, g, CliqueCount<gmsx::SortedSpanSet, HipSetGraph, gmsx::SortedSpanSet>, '
             'CliqueCountVerifier<SortedSet, SortedSetGraph, SortedSet>, k, "HipSet", "HipSetGraph");\n'
             '    BenchmarkKernel(args, g, CliqueCount<gmsx::SortedSpanSet, HipSetRefGraph, gmsx::SortedSpanRef>, '
             'CliqueCountVerifier<RoaringSet, RoaringGraph, RoaringSet>, k, "HipSetRef", "HipSetRefGraph");'),
        drop=[r"\s*BenchmarkKernel\(args, g, CliqueCount<RoaringSet, RoaringGraph, RoaringSet>,[^;]*;",
              r"\s*BenchmarkKernel\(args, g, CliqueCount<SortedSet, SetGraph<SortedSetRef>, SortedSetRef>,[^;]*;",
              r"\s*BenchmarkKernel\(args, g, CliqueCount<SortedSet, SortedSetGraph, SortedSet>,[^;]*;"],
        flags=[]),
    "bron_kerbosch": dict(
        src="gms/algorithms/set_based/maximal_clique_enum/maximal_clique_enum_bron_kerbosch.cc",
        anchor="    runEppstein<SortedSetGraph>(args, g);",
        add="    runEppstein<HipRoaringGraph>(args, g); runEppstein<HipSetGraph>(args, g);",
        drop=[r"\s*runEppstein<RoaringGraph>\(args, g\);", r"\s*runSubGraphs<RoaringGraph>\(args, g\);", r"\s*runEppstein<RobinHoodGraph>\(args, g\);",
              r"\s*runEppstein<SortedSetGraph>\(args, g\);"],
        flags=["-DBK_COUNT"]),
}


def have_ref():
    return os.path.isdir(os.path.join(REF, "gms"))


def patched_source(name, lean):
    d = DRIVERS[name]
    with open(os.path.join(REF, d["src"])) as f:
        text = f.read()
    assert d["anchor"] in text, f"anchor line not found in {d['src']}"
    head, tail = text.split(d["anchor"], 1)
    if name == "k_clique_count":   # the added line goes BEFORE `return 0;`
        text = head + d["add"] + "\n" + d["anchor"] + tail
    else:                          # … or AFTER the last existing flavour
        text = head + d["anchor"] + "\n" + d["add"] + tail
    if lean:
        for pat in d["drop"]:
            text, k = re.subn(pat, "", text, count=1, flags=re.S)
            assert k == 1, pat
    # the one include line of INTEGRATION.md §2, after the driver's own includes
    idx = text.rindex("#include")
    eol = text.index("\n", idx)
    return text[:eol + 1] + "#include <gmsx_gms_glue.hpp>\n" + text[eol + 1:]


def build(name, workdir, lean, exe):
    d = DRIVERS[name]
    src = os.path.join(workdir, name + "_patched.cc")
    with open(src, "w") as f:
        f.write(patched_source(name, lean))
    roaring = os.path.join(ROOT, "oracle", "_ref", "roaring.o")
    assert os.path.exists(roaring), "make -C oracle ref"
    # the reference's flags of record (CMakeLists.txt:10-11,32) with a portable -march; include dir of the driver for "verifier.h"
    cmd = ["g++", "-std=c++17", "-O2", "-march=x86-64-v3", "-fopenmp", "-w", "-DNOPAPIW"] + d["flags"] + \
          ["-I", os.path.join(ROOT, "include"), "-I", REF, "-I", os.path.dirname(os.path.join(REF, d["src"])), src, roaring,
           "-L", LIBDIR, "-lgmsx", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,$ORIGIN/../../gms_amd/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.run(cmd, check=True)
    os.remove(src)  # patched reference text never stays on disk


def build_all(lean=True):
    if not have_ref():
        return False
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        for name in sorted(DRIVERS):
            build(name, td, lean=lean, exe=os.path.join(OUT, name))
    return True


if __name__ == "__main__":
    print("built" if build_all() else "reference tree absent: nothing built")
