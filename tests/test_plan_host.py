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
    if (argc >= 2 && argv[1][0] == 's') {  // s <n_streams> <mode> <has_form> [pin] [n_layers]: skip-lane slots of the form that runs
        printf("%d\n", wn_v3_slots_for(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]) != 0, (argc > 5 && argv[5][0] != '-') ? argv[5] : nullptr, argc > 6 ? atoi(argv[6]) : 50));
        return 0;
    }
    if (argc >= 2 && argv[1][0] == 'r') {  // r <n_streams>: round sizes
        for (int n : wn_v3_round_sizes(atoi(argv[2]), WN_V3_ROUND_STREAMS)) printf("%d\n", n);
        return 0;
    }
    if (argc >= 2 && argv[1][0] == 't') {  // t <nshare> <ngroups>: workgroup id -> (group, member) of the weight-gradient grids
        const unsigned nshare = atoi(argv[2]), ngroups = atoi(argv[3]), blocks = 8u * nshare * ((ngroups + 7u) / 8u);
        for (unsigned id = 0; id < blocks; ++id) {
            unsigned g, m;
            const bool ok = wn_tile_of(id, nshare, ngroups, g, m);
            printf("%u %d %u %u\n", id, ok ? 1 : 0, g, m);
        }
        return 0;
    }
    if (argc >= 2 && argv[1][0] == 'g') {  // g <M> <Ka> <Nb> <tile_nb> <want>: grid of a weight-gradient product
        const WnTnGrid g = wn_tn_grid(atoll(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]));
        printf("%d %d %d %lld %u\n", g.tiles_ka, g.tiles_nb, g.splits, g.rows_per_split, g.blocks);
        return 0;
    }
    if (argc >= 2 && argv[1][0] == 'p') {  // p <R> <D> <S> <E> <n_layers> [max_workgroups]: the compiled shape a model is zero-padded into
        const WnShapeRow rows[] = {{128, 32, 512, 32, 4}, {64, 64, 256, 64, 1}, {32, 32, 256, 64, 1}, {32, 16, 1024, 32, 2}, {64, 32, 256, 64, 2},
                                   {16, 16, 256, 32, 1}, {16, 16, 256, 32, 2}};   // csrc/wn_runtime.hip: wn_v2_table()
        const int NL = atoi(argv[6]), cap = argc > 7 ? atoi(argv[7]) : 252;
        int out[4] = {0, 0, 0, 0};
        const int pick = wn_pad_pick(rows, 7, atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), NL,
                                     [&](int R2, int D2, int S2, int E2) {   // (stand-in for wn_v3_applicable: the chain must fit the CUs)
                                         for (const WnShapeRow& e : rows)
                                             if (e.R == R2 && e.S == S2 && e.DC * e.Pm == D2 && E2 % e.EC == 0) return NL * e.Pm + E2 / e.EC <= cap;
                                         return false;
                                     }, out);
        printf("%d %d %d %d %d\n", pick, out[0], out[1], out[2], out[3]);
        return 0;
    }
    if (argc >= 2 && argv[1][0] == 'v') { printf("%d\n", wn_v4_stream_limit(atoi(argv[2]))); return 0; }   // v <n_stack>: stream limit of the stacked kernel
    if (argc >= 2 && argv[1][0] == 'f') {  // f <layers> <blocks> <L> <out_len>: time geometry of forward() (a | rows | zlo), or the refusal
        const int layers = atoi(argv[2]), blocks = atoi(argv[3]);
        std::vector<int32_t> dil;
        for (int b = 0; b < blocks; ++b) for (int i = 0; i < layers; ++i) dil.push_back(1 << i);
        WnFwdGeom g;
        const std::string why = wn_forward_geometry_host(dil.data(), (int)dil.size(), atoll(argv[4]), atoll(argv[5]), g);
        if (!why.empty()) { printf("REFUSED %s\n", why.c_str()); return 0; }
        for (long long v : g.a) printf("%lld ", v); printf("| ");
        for (long long v : g.rows) printf("%lld ", v); printf("| ");
        for (long long v : g.zlo) printf("%lld ", v); printf("\n");
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


@pytest.mark.parametrize("nshare,ngroups", [(1, 1), (2, 997), (2, 24), (20, 24), (20, 25), (5, 7), (3, 8)])
def test_weight_gradient_tiles_share_an_xcd(harness, nshare, ngroups):
    """wn_tile_of: every (row split, tile) exactly once, the padding rejected, all tiles of a split on ONE XCD (workgroup id % 8) and
    next to each other in dispatch order (ids 8 apart), splits spread round-robin over the XCDs."""
    rows = [tuple(int(x) for x in l.split()) for l in subprocess.check_output([harness, "t", str(nshare), str(ngroups)]).decode().splitlines()]
    assert len(rows) == 8 * nshare * ((ngroups + 7) // 8)
    live = [(g, m, i) for i, ok, g, m in rows if ok]
    assert sorted((g, m) for g, m, _ in live) == [(g, m) for g in range(ngroups) for m in range(nshare)]
    assert all(g >= ngroups for i, ok, g, m in rows if not ok)
    for g in range(ngroups):
        ids = sorted(i for gg, m, i in live if gg == g)
        assert len({i % 8 for i in ids}) == 1 and ids == list(range(ids[0], ids[0] + 8 * nshare, 8))
        assert ids[0] % 8 == g % 8


@pytest.mark.parametrize("M,Ka,Nb,tile_nb,want", [(348320, 256, 256, 256, 512), (512000, 128, 128, 128, 1024), (348320, 512, 1280, 256, 512),
                                                  (348320, 1280, 512, 256, 512), (700, 128, 128, 128, 1024), (64, 32, 64, 128, 1024), (5000, 256, 256, 256, 512)])
def test_weight_gradient_grid(harness, M, Ka, Nb, tile_nb, want):
    """wn_tn_grid: the splits cover every row, are multiples of 32 rows, never shorter than 256 rows unless there is only one, come in
    whole rounds of 8 where there are that many (level XCDs), and the grid has room for every (split, tile)."""
    tka, tnb, splits, rps, blocks = (int(x) for x in subprocess.check_output([harness, "g", str(M), str(Ka), str(Nb), str(tile_nb), str(want)]).decode().split())
    assert tka == -(-Ka // 128) and tnb == -(-Nb // tile_nb)
    assert rps % 32 == 0 and splits * rps >= M and (splits - 1) * rps < M
    assert splits == 1 or rps >= 256
    assert splits * tka * tnb <= max(want, tka * tnb) + 8 * tka * tnb   # about `want` workgroups
    assert blocks == 8 * tka * tnb * -(-splits // 8) and blocks >= splits * tka * tnb
    if M >= 8 * 256 * 2 and want // (tka * tnb) >= 8:
        assert splits % 8 == 0 or splits * rps - M < rps    # whole rounds of 8 (the last split may be cut off by the row count)


@pytest.mark.parametrize("model,n_layers,expect", [
    ((24, 24, 200, 100), 8, (32, 32, 256, 128)),        # the cfg1-shape kernel
    ((40, 20, 256, 250), 6, (64, 64, 256, 256)),        # residual != dilation channels: the unsplit 64-channel kernel (fewest workgroups)
    ((48, 48, 300, 200), 8, (128, 128, 512, 224)),      # only the cfg3-shape kernel holds 300 skip channels next to 48 residual ones
    ((7, 10, 13, 9), 8, (16, 16, 256, 32)),             # (R, D, S, E) of the tests' odd shape
    ((32, 32, 1000, 500), 30, (32, 32, 1024, 512)),     # the train_script.py shape's kernel
    ((32, 32, 256, 256), 10, (32, 32, 256, 256)),       # an exact table shape maps to itself
    ((200, 128, 512, 256), 8, None),                    # wider than any compiled shape: not padded (generic kernel)
    ((32, 32, 2000, 256), 8, None),
])
def test_zero_padding_picks_the_cheapest_shape_that_holds_the_model(harness, model, n_layers, expect):
    out = [int(x) for x in subprocess.check_output([harness, "p"] + [str(x) for x in model] + [str(n_layers)]).decode().split()]
    if expect is None:
        assert out[0] == -1
    else:
        assert out[0] >= 0 and tuple(out[1:]) == expect
        assert all(p >= m for p, m in zip(expect, model))


def test_zero_padding_respects_the_planner(harness):
    """A padded shape the chain cannot be planned for (here: more workgroups than CUs) is skipped for the next one that can."""
    # 60 layers of 64 / 64 / 256: the two-slice 64-channel kernel needs 60 x 2 + 4 workgroups, the unsplit one 60 + 4
    out = [int(x) for x in subprocess.check_output([harness, "p", "64", "64", "256", "256", "60", "100"]).decode().split()]
    assert tuple(out[1:]) == (64, 64, 256, 256) and out[0] == 1
    out = [int(x) for x in subprocess.check_output([harness, "p", "100", "100", "256", "256", "60", "100"]).decode().split()]
    assert out[0] == -1   # only the cfg3-shape kernel holds 100 channels, and 60 x 4 slices do not fit 100 workgroups


def _geometry(harness, layers, blocks, L, out_len):
    out = subprocess.check_output([harness, "f", str(layers), str(blocks), str(L), str(out_len)]).decode().strip()
    if out.startswith("REFUSED"):
        return None
    a, rows, zlo = [[int(v) for v in part.split()] for part in out.split("|")]
    return a, rows, zlo


def test_forward_geometry_of_full_and_short_clips(harness):
    """wn_forward_geometry_host (csrc/wn_plan.h): where every layer's sequence starts once the reference's left zero padding is
    accounted for (wavenet_modules.py:24-27), which rows are computed and which of them read a pad zero as their tap."""
    # a clip of receptive_field + output_length - 1 samples: no returned position sees a pad zero, rows grow by d per layer downwards
    a, rows, zlo = _geometry(harness, 5, 2, 63 + 8 - 1, 8)
    d = [1, 2, 4, 8, 16] * 2
    assert rows[-1] == 8 and all(rows[l] == rows[l + 1] + d[l] for l in range(10)) and rows[0] == 70 and not any(zlo)
    # the golden_v4 cases (made by the real reference): accepted, and some returned rows DO read pad zeros
    for layers, blocks, L, out_len in ((5, 2, 64, 5), (5, 2, 69, 8), (3, 2, 16, 4), (10, 3, 2600, 16), (10, 3, 2751, 6)):
        g = _geometry(harness, layers, blocks, L, out_len)
        assert g is not None, (layers, blocks, L)
        a, rows, zlo = g
        assert any(zlo), (layers, blocks, L)
        assert all(rows[l] <= L - a[l] for l in range(len(rows))) and rows[-1] == out_len
        assert all(a[l + 1] > a[l] for l in range(len(a) - 1))
    # lengths at which the reference has no defined result are refused
    assert _geometry(harness, 5, 2, 41, 5) is None      # the skip path's un-dilation quirk
    assert _geometry(harness, 5, 2, 63, 5) is None      # fewer than output_length positions left
    assert _geometry(harness, 3, 2, 1, 1) is None


@pytest.mark.reference
@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference tree (authoring container)")
def test_forward_geometry_agrees_with_the_live_reference(harness):
    """Every clip length 2 .. rf + output_length + 2 of a small model: the geometry accepts a length exactly when the real reference's
    forward() returns (N * output_length, classes) logits -- and refuses it exactly when the reference raises or returns another shape."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "pytorch-wavenet_amd"))
    import ref_shim
    mdl, _wm, _ad = ref_shim.load()
    for layers, blocks, out_len, N in ((3, 2, 4, 2), (4, 2, 3, 1), (5, 1, 6, 3)):
        m = mdl.WaveNetModel(layers=layers, blocks=blocks, dilation_channels=4, residual_channels=4, skip_channels=8, end_channels=8,
                             classes=16, output_length=out_len)
        for L in range(2, m.receptive_field + out_len + 3):
            x = torch.zeros(N, 16, L)
            x[:, 0, :] = 1.
            try:
                with torch.no_grad():
                    y = m(x)
                ok = tuple(y.shape) == (N * out_len, 16)
            except Exception:   # noqa: BLE001 -- whatever the reference raises there
                ok = False
            assert (_geometry(harness, layers, blocks, L, out_len) is not None) == ok, (layers, blocks, out_len, N, L, ok)


def test_stream_limit_of_the_stacked_kernel(harness):
    """wn_v4_stream_limit: up to how many streams variant 4's short pipeline beats variant 3's (measured on MI355X, profiles/r04_v4_vs_v3_streams.txt)."""
    lim = lambda n: int(subprocess.check_output([harness, "v", str(n)]).decode())
    assert lim(10) == 6 and lim(2) == 2 and lim(15) == 8 and lim(1) == 1 and lim(0) == 1


def test_skip_lane_slot_form_by_stream_count(harness):
    """wn_v3_slots_for: the slot re-use form of cfg3's two-streams-per-item kernel from 96 streams up (2 n_layers - 4: where the measured gain
    starts), never without the two-streams form, never for shapes that have no such kernel, pinnable for tests."""
    run = lambda *a: int(subprocess.check_output([harness, "s"] + [str(x) for x in a]).decode().split()[0])   # noqa: E731
    assert [run(n, 3, 1) for n in (64, 80, 94, 96, 112, 128, 150)] == [0, 0, 0, 4, 4, 4, 4]
    assert run(128, 0, 1) == 0 and run(128, 2, 1) == 0        # one stream per item: no such form
    assert run(128, 3, 0) == 0                                 # a shape without the kernel (everything but cfg3's)
    assert run(64, 3, 1, "4") == 4 and run(128, 3, 1, "0") == 0   # WN_V3_SLOTS pins (tests)
    assert run(6, 3, 1, "4") == 0                              # fewer items than slots: off
    assert run(20, 3, 1, "-", 12) == 4 and run(18, 3, 1, "-", 12) == 0   # the rule follows the depth
