"""Per-hop anatomy of the generation chain from in-kernel wall-clock stamps (wn_profile_next / wn_profile_read).

    python tools/profile_chain.py [cfg3] [n_streams]

Prints, per chain stage: hand-off latency (producer published -> consumer staged), critical compute
(staged -> x' published), tail (published -> step done) and the loop period.  100 MHz stamps (10 ns).
"""
import os
import sys

os.environ.setdefault("WN_TESTING", "1")  # dev tool: WN_V3_MODE / WN_KERNEL pins are honoured

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-wavenet_amd"))
import numpy as np  # noqa: E402

from mi355_wavenet import _abi, engine, synth  # noqa: E402

if os.environ.get("WN_DEV_LIB"):  # dev only: an ablation build of the same source (tools/ablate.sh)
    _abi.PRODUCT_LIB = os.environ["WN_DEV_LIB"]


def main():
    cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cfg = synth.CONFIGS[cfgname]
    W = synth.init_weights(cfg, seed=0)
    eng = engine.Engine(cfg, W, n_streams=ns)
    info = eng.info()
    P, PA, NL = info["layer_split"], info["head_split"], info["n_layers"]
    N = 400 if ns == 1 else 60
    G = max(1, info.get("streams_per_item", 1))   # streams per pipeline item of the layer workgroups (the form that runs: wn_get_info)
    HG = max(1, info.get("head_replicas", 1))     # replicas of the head workgroups (each serves every HG-th stream)
    n_total = ns
    nc = max(1, info.get("n_chains", 1))
    ns = ns // nc          # the stamps are those of the first chain: its streams
    info["n_workgroups"] //= nc
    if nc > 1:
        print("%d chains of %d streams share the CUs; stamps of chain 0" % (nc, ns))
    items = N * ns // G  # pipeline items of a layer workgroup: G streams each
    items_h = N * ns // HG
    u = np.random.RandomState(0).random_sample((n_total, N))
    eng.generate(N, None, temperature=1.0, uniforms=u)  # warm-up
    eng.profile_next(N * ns)
    eng.generate(N, None, temperature=1.0, uniforms=u)
    raw = eng.profile_read(N * ns)
    if ns > 1:
        tt = raw[P:NL * P, items // 4:items - ns // G].astype(np.float64) * 0.01
        print("multi (critical waves): input in registers %.3f us after start, barrier A passed %.3f us, z staged (barrier B) %.3f us, x' published %.3f us" % (
            (tt[:, :, 4] - tt[:, :, 0]).mean(), (tt[:, :, 1] - tt[:, :, 0]).mean(), (tt[:, :, 5] - tt[:, :, 0]).mean(),
            (tt[:, :, 2] - tt[:, :, 0]).mean()))
        if info["kernel_variant"] == 3:
            rw = raw[P:NL * P, items // 4:items - ns // G]
            m40 = (1 << 40) - 1
            c0 = rw[:, :, 0] & m40
            s6, q7 = rw[:, :, 6], rw[:, :, 7]
            s_t0 = ((s6 & m40) - c0) * 0.01
            s_len = (s6 >> 40) * 0.01
            q_t0 = ((q7 & m40) - c0) * 0.01
            q_push = ((q7 >> 40) & 0xfff) * 0.01
            q_dot = ((q7 >> 52) & 0xfff) * 0.01
            print("skip group: passes barrier B %.3f us after the critical group's start of the item (critical: %.3f), its chunk (request, 64-FMA dot, add, publish) "
                  "takes %.3f us; queue group: passes barrier A at %.3f us, push + tap staging %.3f us, tap-0 dot + prefetch %.3f us" % (
                      s_t0.mean(), (tt[:, :, 5] - tt[:, :, 0]).mean(), s_len.mean(), q_t0.mean(), q_push.mean(), q_dot.mean()))
    st = raw.astype(np.float64) * 0.01  # us
    lo, hi = items // 4, items - ns // G  # steady state
    T = st[:, lo:hi, :]
    lay = T[:NL * P].reshape(NL, P, hi - lo, 8)
    head = T[NL * P:]
    Th = st[NL * P:NL * P + PA * HG, items_h // 4:items_h - ns // HG, :]  # the head workgroups' items
    period = np.diff(st[0, lo:hi:ns, 1]).mean() if ns == 1 else np.diff(st[0, lo:hi, 1][::ns // G]).mean()
    print("%s x%d: variant %d P=%d PA=%d workgroups %d; loop period %.2f us/eval (%.0f evals/s per stream, %.0f samples/s total)" % (
        cfgname, ns, info["kernel_variant"], P, PA, info["n_workgroups"], period, 1e6 / period, ns * 1e6 / period))
    if ns > 1:  # pipeline view: who is busy, who waits
        busy = (T[:, :, 3] - T[:, :, 1]).mean(axis=1)
        wait = (T[:, :, 1] - T[:, :, 0]).mean(axis=1)
        per = np.diff(T[:, :, 0], axis=1).mean(axis=1)
        nlw = NL * P
        print("G=%d streams per item; per-item period %.3f us (%.3f us per stream-step).  stage: busy (staged->done) / wait (start->staged), us" % (G, per[:nlw].mean(), per[:nlw].mean() / G))
        for l in list(range(0, NL, max(1, NL // 10))) + [NL - 1]:
            print("  layer %2d: busy %s  wait %s" % (l, np.array2string(busy[l * P:(l + 1) * P], precision=2), np.array2string(wait[l * P:(l + 1) * P], precision=2)))
        print("  head (%d replica(s)): busy %s  wait %s; staged->published %.3f" % (
            HG, np.array2string((Th[:, :, 3] - Th[:, :, 1]).mean(axis=1), precision=2), np.array2string((Th[:, :, 1] - Th[:, :, 0]).mean(axis=1), precision=2),
            (Th[:, :, 2] - Th[:, :, 1]).mean()))
        n_smp = info["n_workgroups"] - nlw - PA * HG
        for j in range(n_smp):
            rows = st[nlw + PA * HG + j, N * ns // 4:N * ns - ns]
            rows = rows[rows[:, 0] > 0]
            if len(rows) > 2:
                print("  sampler %d: waits %.2f us for the logits, samples + publishes in %.2f us; one token every %.2f us" % (
                    j, (rows[:, 1] - rows[:, 0]).mean(), (rows[:, 2] - rows[:, 1]).mean(), np.diff(rows[:, 0]).mean()))
        # cross-workgroup hand-off: x' published by layer l-1 (latest of its P slices) -> layer l has its input in registers
        m40 = (1 << 40) - 1
        rawT = raw[:nlw, lo:hi, :] & m40
        layT = rawT.reshape(NL, P, hi - lo, 8).astype(np.float64) * 0.01
        pub_prev = layT[:-1, :, :, 2].max(axis=1)                      # (NL-1, items)
        got = layT[1:, :, :, 4]                                        # (NL-1, P, items)
        hop = got - pub_prev[:, None, :]
        start_next = layT[1:, :, :, 0]
        early = (start_next <= pub_prev[:, None, :])                   # the consumer was already looking when the data was published
        print("  hand-off x' published -> consumer has it in registers: mean %.3f us (p10 %.3f, p50 %.3f, p90 %.3f); the consumer was already waiting in %.0f%% of the hand-offs; "
              "when it was: %.3f us, when it came later: %.3f us after ITS start" % (
                  hop.mean(), np.percentile(hop, 10), np.percentile(hop, 50), np.percentile(hop, 90), 100.0 * early.mean(),
                  hop[early].mean() if early.any() else float("nan"), (got - start_next)[~early].mean() if (~early).any() else float("nan")))
        by_layer = hop.mean(axis=(1, 2))
        print("  hand-off by consumer layer: %s" % np.array2string(by_layer, precision=2, max_line_width=200))
        if os.environ.get("WN_PROFILE_LAYERS"):  # per-layer table: where the cycle of each stage goes
            L4 = T[:nlw].reshape(NL, P, hi - lo, 8)
            perl = np.diff(L4[:, :, :, 0], axis=2).mean(axis=(1, 2))
            print("  per layer: period | start->input | ->A | ->B | ->published | ->end | skip chunk | queue push, dot   (us)")
            rwl = raw[:nlw, lo:hi].reshape(NL, P, hi - lo, 8)
            for l in range(NL):
                a = L4[l]
                sk = ((rwl[l, :, :, 6] >> 40) * 0.01).mean()
                qp = (((rwl[l, :, :, 7] >> 40) & 0xfff) * 0.01).mean()
                qd = (((rwl[l, :, :, 7] >> 52) & 0xfff) * 0.01).mean()
                print("   L%02d d=%3d: %.3f | %.3f | %.3f | %.3f | %.3f | %.3f | %.3f | %.3f %.3f" % (
                    l, 1 << (l % 10), perl[l], (a[:, :, 4] - a[:, :, 0]).mean(), (a[:, :, 1] - a[:, :, 4]).mean(), (a[:, :, 5] - a[:, :, 1]).mean(),
                    (a[:, :, 2] - a[:, :, 5]).mean(), (a[:, :, 3] - a[:, :, 2]).mean(), sk, qp, qd))
        if info["kernel_variant"] == 3 and n_smp > 0:
            # the tail of the ring, per (stream, evaluation): last layer's barrier B -> head staged -> logits published -> sampler has
            # them -> layer 0's row published -> layer 0 has its input in registers (evaluation e+1)
            nI, n_mine = ns // G, ns // HG
            rf = (raw & m40).astype(np.float64) * 0.01
            seg = []
            for e in range(N // 4, N - 2):
                for s in range(ns):
                    b49 = rf[(NL - 1) * P:NL * P, e * nI + s // G, 5].max()
                    rep = s % HG
                    hrow = rf[nlw + rep * PA:nlw + (rep + 1) * PA, e * n_mine + (s - rep) // HG]
                    smp = rf[nlw + PA * HG + s % n_smp, e * ns + s]
                    l0 = rf[0:P, (e + 1) * nI + s // G, 4].max()
                    l0a = rf[0:P, (e + 1) * nI + s // G, 1].max()
                    seg.append([hrow[:, 1].max() - b49, hrow[:, 2].max() - hrow[:, 1].max(), smp[1] - hrow[:, 2].max(), smp[2] - smp[1], l0 - smp[2], l0a - l0,
                                l0a - b49])
            seg = np.array(seg)
            print("  ring tail per token (us): L%d barrier B -> head staged %.3f | -> logits published %.3f | -> sampler has them %.3f | -> row published %.3f | "
                  "-> L0 input in registers %.3f | -> L0 barrier A %.3f || total %.3f (p50 %.3f, p90 %.3f)" % (
                      NL - 1, *seg.mean(axis=0), np.percentile(seg[:, 6], 50), np.percentile(seg[:, 6], 90)))
        crit = (T[:nlw, :, 2] - T[:nlw, :, 1]).mean()
        print("  layer staged->published %.3f us, published->done %.3f us" % (crit, (T[:nlw, :, 3] - T[:nlw, :, 2]).mean()))
        inp = (T[P:nlw, :, 4] - T[P:nlw, :, 0]).mean()
        if info["kernel_variant"] == 3:
            print("  layers>0: start->input in registers %.3f us, ->barrier A passed %.3f us, ->barrier B passed %.3f us, ->x' published %.3f us, ->requests issued %.3f us" % (
                inp, (T[P:nlw, :, 1] - T[P:nlw, :, 4]).mean(), (T[P:nlw, :, 5] - T[P:nlw, :, 1]).mean(), (T[P:nlw, :, 2] - T[P:nlw, :, 5]).mean(),
                (T[P:nlw, :, 3] - T[P:nlw, :, 2]).mean()))
        else:
            print("  layers>0: start->input in registers %.3f us, ->barrier passed %.3f us; request misses %.0f%% of items" % (
                inp, (T[P:nlw, :, 1] - T[P:nlw, :, 4]).mean(), 100.0 * (st[P:nlw, hi - 1, 5] - st[P:nlw, lo, 5]).mean() / 0.01 / (hi - 1 - lo)))
        eng.close()
        return
    # layer hops
    hop = lay[1:, :, :, 1] - lay[:-1, :, :, 2].max(axis=1)[:, None, :]   # staged(l,c) - max_cc published(l-1,cc)
    crit = lay[:, :, :, 2] - lay[:, :, :, 1]
    tail = lay[:, :, :, 3] - lay[:, :, :, 2]
    wait = lay[:, :, :, 1] - lay[:, :, :, 0]
    print("layer hand-off (published -> staged): mean %.3f us  p50 %.3f  p95 %.3f  [max over the %d lanes: %.3f]" % (
        hop.mean(), np.median(hop), np.percentile(hop, 95), P, hop.max(axis=1).mean()))
    print("layer critical compute (staged -> x' published): mean %.3f us (L0: %.3f)" % (crit[1:].mean(), crit[0].mean()))
    print("  of which: fg matvec+reduce %.3f us, gating+barrier %.3f us, residual+publish %.3f us" % (
        (lay[1:, :, :, 4] - lay[1:, :, :, 1]).mean(), (lay[1:, :, :, 5] - lay[1:, :, :, 4]).mean(),
        (lay[1:, :, :, 2] - lay[1:, :, :, 5]).mean()))
    print("layer tail (published -> done): mean %.3f us;  poll wait inside the step: %.3f us" % (tail.mean(), wait[1:].mean()))
    per_layer = (lay[1:, :, :, 2].max(axis=1) - lay[:-1, :, :, 2].max(axis=1)).mean(axis=1)
    print("published(l) - published(l-1): mean %.3f us; by layer %s" % (per_layer.mean(), np.array2string(per_layer, precision=2)))
    # head: staged - last layer's done (skip lane is published inside the tail)
    h_in = head[:, :, 1] - lay[-1, :, :, 2].max(axis=0)[None, :]
    h_c = head[:, :, 2] - head[:, :, 1]
    print("head: last layer x' published -> head staged %.3f us; head compute %.3f us" % (h_in.mean(), h_c.mean()))
    # L0: staged(e+1) - head published(e)
    l0 = lay[0, :, ns:, 1] - head[:, :-ns, 2].max(axis=0)[None, :]
    print("L0: head published -> L0 input staged (logit hop + sampler + start_conv) %.3f us" % l0.mean())
    eng.close()


if __name__ == "__main__":
    main()
