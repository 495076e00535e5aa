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
    if (argc >= 2 && argv[1][0] == 'm') {  // m <n_streams> [pin]: the form of the wave-specialised kernel
        printf("%d\n", wn_v3_mode_for(atoi(argv[2]), (argc > 3 && argv[3][0] != '-') ? argv[3] : nullptr, argc > 4 ? atoi(argv[4]) : 50));
        return 0;
    }
    if (argc >= 2 && argv[1][0] == 'r') {  // r <n_streams>: round sizes
        for (int n : wn_v3_round_sizes(atoi(argv[2]), WN_V3_ROUND_STREAMS)) printf("%d\n", n);
        return 0;
    }
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
    n_used = len(set(order))
    assert (n_used - 1) * 32 < n_wg  # no more XCDs than needed
    # the token moves down through the XCDs in order; the LAST layer sits next to the head on XCD 0 when the chain spans several
    # XCDs and XCD 0 has room for two layers (its skip lanes are the widest hand-off of the ring): the ring still crosses exactly
    # n_used boundaries per trip
    if n_used > 1 and order[-1] == 0:
        assert order[:-1] == sorted(order[:-1]) and order[-2] == n_used - 1
    else:
        assert order == sorted(order)
    crossings = sum(1 for a, b in zip(order, order[1:]) if a != b) + (1 if order[-1] != 0 else 0)  # ... + last layer -> head on XCD 0
    assert crossings == (n_used if n_used > 1 else 0)
    if n_used > 1 and (32 - heads - n_smp) // P >= 2:
        assert order[-1] == 0


def test_placement_refuses_what_does_not_fit(harness):
    assert run(harness, 50, 4, 40, 4)[0] == 0   # head + samplers must fit one XCD
    assert run(harness, 80, 4, 8, 4)[0] == 0    # 332 workgroups > 256 CUs


def test_form_of_the_wave_specialised_kernel(harness):
    """bit 0 (two streams per layer item) needs an even stream count, the throughput form starts at n_layers + 6 streams (cfg3: 56),
    WN_V3_MODE pins."""
    def mode(n, pin=None, n_layers=50):
        return int(subprocess.check_output([harness, "m", str(n), pin if pin is not None else "-", str(n_layers)]).decode())
    assert [mode(n) for n in (1, 2, 16, 48, 55)] == [0, 0, 0, 0, 0]
    assert [mode(n) for n in (56, 64, 128)] == [3, 3, 3]
    assert [mode(n) for n in (57, 63, 129)] == [2, 2, 2]          # odd: one stream per item, two head replicas
    assert [mode(6, p) for p in ("0", "1", "2", "3")] == [0, 1, 2, 3]
    assert [mode(7, p) for p in ("0", "1", "2", "3")] == [0, 0, 2, 2]
    assert mode(1, "3") == 0 and mode(64, "0") == 0 and mode(64, "7") == 3 and mode(64, "12") == 3  # junk is ignored
    # the threshold follows the ring's length: n_layers + 6 streams (cfg2: 30 layers, cfg1: 10)
    assert [mode(n, None, 30) for n in (32, 35, 36, 64)] == [0, 0, 3, 3]
    assert [mode(n, None, 10) for n in (14, 16, 17, 64)] == [0, 3, 2, 3]


@pytest.mark.parametrize("ns", [129, 130, 170, 192, 255, 256, 257, 301, 381, 383, 384, 385, 512, 1000])
def test_round_sizes(harness, ns):
    sizes = [int(x) for x in subprocess.check_output([harness, "r", str(ns)]).decode().split()]
    assert sum(sizes) == ns and len(sizes) == -(-ns // 128)
    assert max(sizes) <= 128 and min(sizes) >= 1
    assert max(sizes) - min(sizes) <= 3
    assert sum(1 for n in sizes if n % 2) <= 1  # at most the last round is odd
    assert all(n % 2 == 0 for n in sizes[:-1])
