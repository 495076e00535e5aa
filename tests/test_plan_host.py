"""CPU: the host-side placement of the chain's workgroups (csrc/wn_plan.h, plain C++) compiled with g++ and checked through a
small harness: every chain position exactly once, the P slices of a layer on ONE XCD, head (replicas) and samplers on XCD 0,
as few XCDs as hold the chain -- for the geometries wn_create builds (cfg3: 50 layers x 4 slices, 8 head workgroups or two
replicas of them, 4 samplers)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pytorch-wavenet_amd", "csrc")

HARNESS = r"""
#include "wn_plan.h"
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv) {
    const int NL = atoi(argv[1]), P = atoi(argv[2]), heads = atoi(argv[3]), n_smp = atoi(argv[4]);
    std::vector<int32_t> m;
    int nb = 0;
    const bool ok = wn_make_wg_map_layers(NL, P, heads, n_smp, 8, 32, m, &nb);
    printf("%d %d\n", ok ? 1 : 0, nb);
    if (ok) for (int b = 0; b < nb; ++b) printf("%d\n", m[b]);
    return 0;
}
"""


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    d = tmp_path_factory.mktemp("plan")
    src = d / "plan_harness.cpp"
    src.write_text(HARNESS)
    exe = d / "plan_harness"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", CSRC, str(src), "-o", str(exe)])
    return str(exe)


def run(exe, NL, P, heads, n_smp):
    out = subprocess.check_output([exe, str(NL), str(P), str(heads), str(n_smp)]).decode().split()
    ok, nb = int(out[0]), int(out[1])
    return ok, nb, [int(x) for x in out[2:]]


@pytest.mark.parametrize("NL,P,heads,n_smp", [(50, 4, 8, 4), (50, 4, 16, 4), (50, 4, 16, 8), (6, 4, 16, 4), (30, 1, 4, 0), (50, 4, 8, 1)])
def test_layer_aligned_placement(harness, NL, P, heads, n_smp):
    ok, nb, m = run(harness, NL, P, heads, n_smp)
    assert ok and nb == len(m) and nb % 8 == 0 and nb <= 256
    n_wg = NL * P + heads + n_smp
    used = [w for w in m if w >= 0]
    assert sorted(used) == list(range(n_wg))  # every chain position exactly once, bystanders are -1
    xcd = {w: b % 8 for b, w in enumerate(m) if w >= 0}  # block b lands on XCD b % 8
    for l in range(NL):
        assert len({xcd[l * P + c] for c in range(P)}) == 1, l  # a layer's slices share an L2
    assert {xcd[NL * P + h] for h in range(heads + n_smp)} == {0}  # head (replicas) and samplers next to layer 0
    assert xcd[0] == 0
    per_xcd = [sum(1 for w in used if xcd[w] == x) for x in range(8)]
    assert max(per_xcd) <= 32
    order = [xcd[l * P] for l in range(NL)]
    assert order == sorted(order)  # the token crosses an XCD boundary at most 7 times on its way down
    n_used = len(set(order))
    assert (n_used - 1) * 32 < n_wg  # no more XCDs than needed


def test_placement_refuses_what_does_not_fit(harness):
    assert run(harness, 50, 4, 40, 4)[0] == 0   # head + samplers must fit one XCD
    assert run(harness, 80, 4, 8, 4)[0] == 0    # 332 workgroups > 256 CUs
