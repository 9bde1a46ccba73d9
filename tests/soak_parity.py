"""One-off soak (not collected by pytest; run as `python tests/soak_parity.py [draws] [seed]` on a GPU box): random
(graph, shape, dropout, segment length, kernel-choice switches) draws of one GAT layer on both paths against the
CPU oracle at the bar of tests/test_gpu_parity.py.  Wider than the seeded sweeps in the suite: node counts up to a few
thousand (around the 32-row tiles and the 4096-row thresholds of the kernel choices), every channel count 1..70,
5-neighbour / hub / self-loop-only graphs, the fused-tile and unpadded-row switches at random.  A draw beyond the bar
is set against the fp64 oracle: if the HIP result is as close to fp64 as the fp32 oracle is, if a logit sits within
2e-7 of LeakyReLU's kink and only that head's slices of the gradients are off, or if the reference is an exactly-zero
sum of cancelling terms, fp32 explains it; anything else is UNEXPLAINED and makes the exit status 1."""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cases  # noqa: E402
from oracle import gat_oracle as O  # noqa: E402
from test_gpu_parity import close, run_layer  # noqa: E402


def draw_graph(rng, n, trial):
    from gat_amd import synth
    kind = rng.choice(["knn", "knn", "multi", "multi", "loops", "deg40"])
    g = torch.Generator().manual_seed(trial)
    if kind == "knn":
        return synth.out_degree_k_graph(n, rng.choice([1, 5, 5, 9]), seed=trial), kind
    if kind == "loops":
        return torch.zeros(2, 0, dtype=torch.long), kind
    if kind == "deg40":
        return torch.stack([torch.randint(0, n, (40 * n,), generator=g), torch.arange(n).repeat_interleave(40)]), kind
    e = rng.choice([1, n, 5 * n, 20 * n])
    ei = torch.randint(0, n, (2, e), generator=g)
    if e > 4:
        ei[1, : e // 3] = ei[1, 0]
        ei[0, e // 3: e // 2] = ei[0, -1]
    return ei, kind


def off_heads(name, got, ref32, r64, sc, h, c):
    """Heads whose slice of a tensor is further from fp64 than the bar and than 4x the fp32 oracle's distance.
    d_x mixes the heads (every head is returned as off only if the tensor has no head axis to look at)."""
    g = got.detach().cpu().double().view_as(r64)
    o = ref32.detach().cpu().double().view_as(r64)
    if name == "x" or r64.numel() % (h * c):
        return set()
    if name == "weight":
        shape = (h, -1)                       # [H*C, F]: rows of a head are contiguous
    elif name == "out":
        g, o, r64 = g.t().contiguous(), o.t().contiguous(), r64.t().contiguous()   # [H*C, N]
        shape = (h, -1)
    else:
        shape = (h, -1)                       # [1, H, C] / [H*C]
    eg = (g - r64).abs().reshape(shape).max(1).values / sc
    eo = (o - r64).abs().reshape(shape).max(1).values / sc
    return {int(k) for k in torch.nonzero((eg > 1e-5) & (eg > 4 * eo)).flatten()}


def main():
    from gat_amd import _lib, ops
    draws = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
    cuda = torch.device("cuda:0")
    only = {int(t) for t in os.environ.get("SOAK_ONLY", "").split(",") if t}   # replay these draws, with a diagnosis
    failed, done, explained = [], 0, 0
    for trial in range(draws):
        n = rng.choice([1, 2, 31, 32, 33, 200, 777, 1500, 3001, 4095, 4096, 4200, 6000])
        h = rng.choice([1, 2, 3, 4, 5, 6, 7, 8])
        c = rng.randint(1, 70)
        inagg = rng.random() < 0.6
        f = rng.choice([1, 3, 4, 12, 16, 25, 25, 29, 32]) if inagg else rng.choice([1, 3, 16, 25, 100, 240, 264])
        if inagg and not ops.inagg_supported(f, h, c):
            continue
        if h * ((c + 3) // 4 * 4) > 1024:
            continue
        ei, kind = draw_graph(rng, n, trial)
        if kind == "deg40" and n > 2000:
            continue
        p = rng.choice([0.0, 0.0, 0.3, 0.425])
        seg = rng.choice([64, 128, 512])
        opts = {"GATMI_FUSED": rng.choice([-1, -1, 0, 9, 15, 1]), "GATMI_GCOMPACT": rng.choice([-1, 0, 2]),
                "GATMI_NARROW": rng.choice([-1, 0])}
        if only and trial not in only:
            continue
        inputs = cases.layer_inputs(n, f, h, c, seed=trial)
        x, w, a_s, a_d, b, go = inputs
        ne = O.build_csr(ei, n).num_edges
        mask = cases.keep_mask(ne, h, p, seed=trial) if p > 0 else None
        ref_out, ref_g = O.gatconv_grads_autograd(x, ei, w, a_s, a_d, b, h, c, go, dropout=p, mask=mask,
                                                  training=p > 0)
        tag = f"trial {trial}: {kind} n={n} e={ei.size(1)} f={f} h={h} c={c} p={p} seg={seg} inagg={inagg} {opts}"
        for k, v in opts.items():
            _lib.set_option(k, v)
        try:
            out, gr = run_layer(cuda, ei, n, f, h, c, p, mask, seg, inputs, inagg=inagg,
                                stream=bool(trial & 1) if inagg else None)
            close(out, ref_out, tag + " out")
            summand = float(go.abs().max() * (x @ w.t()).abs().max())
            for k in gr:
                close(gr[k], ref_g[k], f"{tag} d_{k}", floor=summand if k.startswith("att") else 0.0)
        except AssertionError as err:
            print("FAIL", tag, str(err).splitlines()[0][:300], flush=True)
            if True:
                # how far the HIP result and the fp32 oracle each are from the fp64 oracle, in units of the
                # tensor's largest magnitude: a LeakyReLU slope taken on the other side of zero shows as an
                # error of the same size in BOTH, a kernel bug only in the first
                o64, g64 = O.gatconv_grads_autograd(x.double(), ei, w.double(), a_s.double(), a_d.double(), b.double(),
                                                    h, c, go.double(), dropout=p, mask=mask, training=p > 0)
                got = dict(gr, out=out)
                ref32 = dict(ref_g, out=ref_out)
                ref64 = dict(g64, out=o64)
                # the logits u = a_src[j] + a_dst[i] closest to LeakyReLU's kink, in fp64 and as the fp32 oracle
                # rounds them: the input-side path forms a_src as <x, W^T att> instead of <W x, att>
                hf = (x.double() @ w.double().t()).view(n, h, c)
                asn, adn = (hf * a_s.double().view(1, h, c)).sum(-1), (hf * a_d.double().view(1, h, c)).sum(-1)
                csr = O.build_csr(ei, n)
                src = torch.as_tensor(csr.col, dtype=torch.long)
                dst = torch.repeat_interleave(torch.arange(n), torch.as_tensor(csr.rowptr[1:] - csr.rowptr[:-1], dtype=torch.long))
                u = asn[src] + adn[dst]
                scale_u = float(asn.abs().max() + adn.abs().max())
                near = u.abs().flatten().sort().values[:3] / scale_u
                kink_heads = {int(k) for k in torch.nonzero(u.abs().min(0).values / scale_u < 2e-7).flatten()}
                print(f"    logits nearest to zero, relative to max|a_src| + max|a_dst| = {scale_u:.2e}: "
                      + ", ".join(f"{float(v):.1e}" for v in near), flush=True)
                verdicts = set()
                for k in got:
                    r64 = ref64[k].detach().cpu().double()
                    zero_ref = float(r64.abs().max()) == 0.0
                    sc = max(float(r64.abs().max()), summand if k.startswith("att") else 0.0, 1e-30)
                    e_hip = float((got[k].detach().cpu().double().view_as(r64) - r64).abs().max()) / sc
                    e_o32 = float((ref32[k].detach().cpu().double().view_as(r64) - r64).abs().max()) / sc
                    print(f"    {k:8s} |hip - fp64| {e_hip:.2e}   |oracle32 - fp64| {e_o32:.2e}   scale {sc:.3e}", flush=True)
                    if e_hip <= 1e-5 or e_hip <= 4 * e_o32:
                        continue                             # as close to fp64 as the fp32 oracle is (x4): the oracle's own rounding
                    if zero_ref and e_hip <= 1e-5 * max(1.0, (csr.num_edges * h) ** 0.5 / 16):
                        verdicts.add("an exactly-zero sum of cancelling terms: rounding noise grows with sqrt(terms)")
                    elif kink_heads and off_heads(k, got[k], ref32[k], r64, sc, h, c) <= kink_heads:
                        verdicts.add(f"a logit of head {sorted(kink_heads)} within 2e-7 of LeakyReLU's kink takes the other "
                                     "slope, and only that head's slices are off (DESIGN.md section 6)")
                    else:
                        verdicts.add("UNEXPLAINED")
                print("    ->", "; ".join(sorted(verdicts)) or "the fp32 oracle is the one further from fp64", flush=True)
                if "UNEXPLAINED" in verdicts:
                    failed.append(tag)
                else:
                    explained += 1
        finally:
            for k in opts:
                _lib.set_option(k, -1)
        done += 1
        if done % 50 == 0:
            print(f"{done} draws, {len(failed)} failed", flush=True)
    print(f"soak: {done} draws, {explained} beyond the bar and explained by fp32 itself, {len(failed)} unexplained")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
