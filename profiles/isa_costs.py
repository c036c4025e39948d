#!/usr/bin/env python3
"""VALU instructions per unit of the shader's own arithmetic, read off the ISA (no GPU needed).

    python profiles/isa_costs.py [out.json]         (default profiles/r06/isa_costs.json)

profiles/isa_costs.hip wraps each stage of the per-pixel path -- the product's inline functions, unchanged -- in a
kernel that runs it once or twice; the difference of the two instances' VALU counts is one repetition of the stage.
bench.py turns these constants and the frame's work counters into `roofline.algorithmic_ops`:

    lane_ops = c_node Nv + c_tri_distance Tt + c_tri_barycentric Th + c_setup Tr + c_shade H + c_env E + c_pixel S

(Nv node visits, Tt triangle tests, Th tests that reach the barycentric part -- not counted by the kernels, bounded below
by the shaded hits H, which is what is used --, Tr traversals, E environment lookups, S samples), divided by 64 lanes:
the wave-instructions a frame would take if every lane of every instruction did arithmetic the shader asks for and
nothing else was issued.  Unlike the issued-instruction fraction it goes DOWN when bookkeeping grows."""
import json
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_count import classify, disassemble, demangle   # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "shader-ray_amd")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-fhip-fp32-correctly-rounded-divide-sqrt", f"-I{ROOT}/include", f"-I{PKG}/csrc"]


def measure():
    with tempfile.TemporaryDirectory() as tmp:
        obj = os.path.join(tmp, "isa_costs.o")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-c", os.path.join(ROOT, "profiles", "isa_costs.hip"), "-o", obj], check=True)
        kernels = disassemble(obj)
    names = [n for n in kernels if not n.endswith(".kd")]
    valu = {}
    for name, pretty in zip(names, demangle(names)):
        short = pretty.replace("void ", "").split("(")[0]
        valu[short] = sum(1 for op in kernels[name] if classify(op) == "valu")
    stage = lambda k: valu[f"{k}<2>"] - valu[f"{k}<1>"]   # noqa: E731
    full = stage("cost_triangle_full")
    dist = stage("cost_triangle_distance")
    return {
        "unit": "VALU instructions per lane per unit of work (static count of the whole stage, gfx950, the product's flags)",
        "c_node": stage("cost_node"),
        "c_node_exact_quotients": stage("cost_node_exact"),
        "c_tri_distance": dist,
        "c_tri_barycentric": full - dist,
        "c_tri_full": full,
        "c_setup": stage("cost_traversal_setup"),
        "c_shade": stage("cost_shade"),
        "c_env": stage("cost_environment"),
        "c_pixel": stage("cost_primary_and_tonemap"),
        "raw_valu_counts": valu,
        "method": "profiles/isa_costs.hip, VALU count of the REPS = 2 instance minus the REPS = 1 instance of each stage",
    }


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06", "isa_costs.json")
    costs = measure()
    # keyed like the counter file: the costs are those of the stages as THIS build's headers compile them (bench.py: keyed_input)
    from buildhash import kernel_source_hash
    costs["build_hash"] = kernel_source_hash()
    json.dump(costs, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in costs.items() if k.startswith("c_")}))
