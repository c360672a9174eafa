#!/usr/bin/env python3
"""What exactly is wrong in a wrong tile of conv3x3_wino4_f32?  (the round-5 run-to-run differences.)

  capture N OUT.npz   (GPU) N launches of the plain kernel on the FPN-P2 shape (batch 2, 256 x 256, 256 -> 256 channels, no
                      activation: the output is LINEAR in every operand), both outputs; every launch compared with the first, bit
                      for bit; of every differing launch the list of differing tiles and, for up to MAX_TILES of them, the tile
                      as computed (NHWC and k-blocked) and as the first launch computed it. Run it on the build under test:
                      MRCNN_LIB=<variant library> (maskrcnn_amd/build.py --variant).
  analyze IN.npz      (CPU, fp64) regenerates the seeded operands, restates the kernel's transforms (B^T, G, A^T at the points
                      0, +-3/4, +-3/2, inf) and decomposes every captured difference D = bad - first:
                        D[pos, i, j, n] = sum_c A^T[i, xi_c] A^T[j, nu_c] dM_c[pos, n]         (c = component xi * 6 + nu)
                      1. which components carry it (one component / one wave's 3 x 3 quadrant / more), which positions, channels;
                      2. per component, is dM_c = V_kt,c (U' - U_kt,c) for ONE k tile kt (rank <= 4, column space = the k tile's
                         transformed input) — a wrong B operand (U) of that k tile — and if so what U' was (zero, another k
                         tile's, another component's);
                      3. or dM_c = (V' - V_kt,c) U_kt,c (row space = the k tile's U) — a wrong A operand / raw input;
                      4. or neither: the accumulated M_c itself replaced (exchange-buffer bytes overwritten) — then M_c + dM_c is
                         compared with what the overwriting bytes could be.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

B, H, W, CIN, COUT = 2, 256, 256, 256, 256
MAX_TILES = 110
A_, B_ = 0.75, 1.5


def operands():
    import torch
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, W, CIN, generator=g)
    w = torch.randn(COUT, 3, 3, CIN, generator=g) * 0.02
    shift = torch.randn(COUT, generator=g)
    return x, w, shift


def capture(n, out_path):
    import torch
    from maskrcnn_amd import ops, _lib
    dev = torch.device("cuda:0")
    x, w, shift = operands()
    xk = ops.nhwc_to_kblocked(x.to(dev))
    u4 = ops.winograd4_weights(w.to(dev))
    shift = shift.to(dev)
    first = firstk = None
    events, tiles_bad, tiles_badk, tiles_first, tile_ids = [], [], [], [], []
    for it in range(n):
        y, yk = ops.conv3x3_winograd4(xk, u4, None, shift, False, None, "both")
        if first is None:
            first, firstk = y.clone(), yk.clone()
            continue
        if torch.equal(y, first) and torch.equal(yk, firstk):
            continue
        d = (y != first)
        t = d.view(B, 16, 16, 8, 32, 4, 64).permute(0, 1, 3, 5, 2, 4, 6).reshape(B, 16, 8, 4, -1).sum(-1)
        units = [(list(ix), int(t[tuple(ix)])) for ix in t.nonzero().tolist()]
        # k-blocked twin [Cout/8, B, H, W, 8] -> NHWC view, compared with the NHWC output of the SAME launch
        ykn = yk.permute(1, 2, 3, 0, 4).reshape(B, H, W, COUT)
        same_both = bool(torch.equal(ykn, y))
        dk = (yk != firstk)
        events.append({"launch": it, "elements": int(d.sum()), "elements_kblocked": int(dk.sum()),
                       "nhwc_equals_kblocked_in_this_launch": same_both,
                       "max_abs_diff": float((y - first).abs().max()), "tiles": units})
        for (b, ty, tx, nt), cnt in units:
            if len(tile_ids) >= MAX_TILES:
                break
            sl = (b, slice(16 * ty, 16 * ty + 16), slice(32 * tx, 32 * tx + 32), slice(64 * nt, 64 * nt + 64))
            tiles_bad.append(y[sl].cpu().numpy())
            tiles_badk.append(ykn[sl].cpu().numpy())
            tiles_first.append(first[sl].cpu().numpy())
            tile_ids.append([it, b, ty, tx, nt, cnt])
        print(json.dumps({k: v for k, v in events[-1].items() if k != "tiles"} | {"n_tiles": len(units)}), flush=True)
    np.savez_compressed(out_path, tile_ids=np.array(tile_ids, dtype=np.int64).reshape(-1, 6),
                        bad=np.array(tiles_bad, dtype=np.float32).reshape(-1, 16, 32, 64),
                        badk=np.array(tiles_badk, dtype=np.float32).reshape(-1, 16, 32, 64),
                        first=np.array(tiles_first, dtype=np.float32).reshape(-1, 16, 32, 64),
                        events=json.dumps(events), launches=n, lib=_lib.LIB_PATH)
    print(json.dumps({"launches": n, "differed": len(events), "tiles_saved": len(tile_ids), "lib": _lib.LIB_PATH}))


# ---- the kernel's transforms, restated (conv_wino4.hip: bt3, at4p, wino4_weights_kernel)
def bt_matrix():
    a, b = A_, B_
    a2, b2 = a * a, b * b
    return np.array([[a2 * b2, 0, -(a2 + b2), 0, 1, 0],
                     [0, -a * b2, -b2, a, 1, 0],
                     [0, a * b2, -b2, -a, 1, 0],
                     [0, -a2 * b, -a2, b, 1, 0],
                     [0, a2 * b, -a2, -b, 1, 0],
                     [0, a2 * b2, 0, -(a2 + b2), 0, 1]], dtype=np.float64)


def g_matrix():
    pts = [0.0, A_, -A_, B_, -B_]
    G = np.zeros((6, 3))
    for j, p in enumerate(pts):
        nj = np.prod([p - q for l, q in enumerate(pts) if l != j])
        G[j] = [1.0 / nj, p / nj, p * p / nj]
    G[5] = [0, 0, 1]
    return G


def at_matrix():
    a, b = A_, B_
    return np.array([[1, 1, 1, 1, 1, 0],
                     [0, a, -a, b, -b, 0],
                     [0, a * a, a * a, b * b, b * b, 0],
                     [0, a ** 3, -a ** 3, b ** 3, -b ** 3, 1]], dtype=np.float64)


class Model:
    def __init__(self):
        x, w, shift = operands()
        self.x = x.numpy().astype(np.float64)
        self.w = w.numpy().astype(np.float64)          # [Cout, 3, 3, Cin]
        self.shift = shift.numpy().astype(np.float64)
        self.BT, self.G, self.AT = bt_matrix(), g_matrix(), at_matrix()
        # U[xi, nu, cout, cin] = G g G^T
        self.U = np.einsum("iy,nyxc,jx->ijnc", self.G, self.w, self.G)
        # P[16, 36]: D[i * 4 + j] = sum_c P[.., c] dM_c
        self.P = np.einsum("ix,jn->ijxn", self.AT, self.AT).reshape(16, 36)

    def tile(self, b, ty, tx, nt):
        """V [64 kt, 36, 32 pos, 4], U [64 kt, 36, 4, 64], M [36, 32, 64] of one workgroup tile."""
        xp = np.zeros((18 + 2, 34 + 2, CIN))
        y0, x0 = 16 * ty - 1, 32 * tx - 1
        ys, xs = np.arange(y0, y0 + 18), np.arange(x0, x0 + 34)
        oky, okx = (ys >= 0) & (ys < H), (xs >= 0) & (xs < W)
        reg = np.zeros((18, 34, CIN))
        reg[np.ix_(oky, okx)] = self.x[b][np.ix_(ys[oky], xs[okx])]
        V = np.zeros((32, 6, 6, CIN))
        for pos in range(32):
            py, px = pos >> 3, pos & 7
            d = reg[4 * py:4 * py + 6, 4 * px:4 * px + 6]           # [6, 6, Cin]
            V[pos] = np.einsum("ir,rcC,jc->ijC", self.BT, d, self.BT)
        V = V.reshape(32, 36, 64, 4).transpose(2, 1, 0, 3)           # [kt, comp, pos, 4]
        U = self.U[:, :, 64 * nt:64 * nt + 64, :].reshape(36, 64, 64, 4).transpose(2, 0, 3, 1)   # [kt, comp, 4, n]
        M = np.einsum("kcpi,kcin->cpn", V, U)
        return V, U, M

    def output(self, M, nt):
        """[16 rows, 32 cols, 64] of the tile from M [36, 32, 64]."""
        Y = np.einsum("dc,cpn->pdn", self.P, M).reshape(4, 8, 4, 4, 64)     # [py, px, i, j, n]
        return Y.transpose(0, 2, 1, 3, 4).reshape(16, 32, 64) + self.shift[64 * nt:64 * nt + 64]


def rel(res, ref):
    return float(np.linalg.norm(res) / max(np.linalg.norm(ref), 1e-300))


def analyze(path):
    z = np.load(path, allow_pickle=False)
    ids, bad, badk, first = z["tile_ids"], z["bad"], z["badk"], z["first"]
    events = json.loads(str(z["events"]))
    print(json.dumps({"lib": str(z["lib"]), "launches": int(z["launches"]), "events": len(events), "tiles": len(ids)}))
    for e in events:
        its = sorted({(4 * (b * 4 + ty // 4) + 0, ty % 4) for (b, ty, tx, nt), _ in e["tiles"]})
        print(json.dumps({"launch": e["launch"], "elements": e["elements"], "kblocked_elements": e["elements_kblocked"],
                          "nhwc==kblocked": e["nhwc_equals_kblocked_in_this_launch"], "max_abs_diff": round(e["max_abs_diff"], 4),
                          "n_tiles": len(e["tiles"]),
                          "persistent_loop_iterations": sorted({ty % 4 for (b, ty, tx, nt), _ in e["tiles"]}),
                          "counts": sorted({c for _, c in e["tiles"]})}))
    m = Model()
    reports = []
    quads = [[(3 * qa + i) * 6 + 3 * qb + j for i in range(3) for j in range(3)] for qa in range(2) for qb in range(2)]
    for t in range(len(ids)):
        it, b, ty, tx, nt, cnt = [int(v) for v in ids[t]]
        D = bad[t].astype(np.float64) - first[t].astype(np.float64)            # [16, 32, 64]
        rep = {"launch": it, "tile": [b, ty, tx, nt], "elements": cnt, "max_abs_diff": float(np.abs(D).max()),
               "kblocked_same_as_nhwc": bool(np.array_equal(bad[t], badk[t]))}
        V, U, M = m.tile(b, ty, tx, nt)
        rep["model_vs_first_max_abs"] = float(np.abs(m.output(M, nt) - first[t]).max())
        Dp = D.reshape(4, 4, 8, 4, 64).transpose(0, 2, 1, 3, 4).reshape(32, 16, 64)       # [pos, i * 4 + j, n]
        pos_bad = [int(p) for p in np.nonzero(np.abs(Dp).max(axis=(1, 2)) > 0)[0]]
        ch_bad = np.nonzero(np.abs(Dp).max(axis=(0, 1)) > 0)[0]
        rep["positions"] = pos_bad if len(pos_bad) < 32 else "all 32"
        rep["rounds"] = sorted({p >> 3 for p in pos_bad})
        rep["channels"] = f"{len(ch_bad)} of 64" + ("" if len(ch_bad) in (0, 64) else f" [{int(ch_bad.min())}..{int(ch_bad.max())}]")
        # 1. support: single component, one quadrant
        single = []
        for c in range(36):
            pc = m.P[:, c]
            coef = np.einsum("d,pdn->pn", pc, Dp) / (pc @ pc)
            single.append(rel(Dp - coef[:, None, :] * pc[None, :, None], Dp))
        best_c = int(np.argmin(single))
        rep["single_component_fit"] = {"component": [best_c // 6, best_c % 6], "residual": round(single[best_c], 6)}
        qres = []
        for q in range(4):
            Pq = m.P[:, quads[q]]                                                # [16, 9]
            sol, *_ = np.linalg.lstsq(Pq, Dp.transpose(1, 0, 2).reshape(16, -1), rcond=None)
            qres.append(rel(Dp.transpose(1, 0, 2).reshape(16, -1) - Pq @ sol, Dp))
        bq = int(np.argmin(qres))
        rep["quadrant_fit"] = {"wave": bq, "residuals": [round(r, 6) for r in qres]}
        support = None
        if single[best_c] < 1e-4:
            support = [best_c]
        elif qres[bq] < 1e-4:
            support = quads[bq]
        if support is not None:
            Ps = m.P[:, support]
            sol, *_ = np.linalg.lstsq(Ps, Dp.transpose(1, 0, 2).reshape(16, -1), rcond=None)
            dM = sol.reshape(len(support), 32, 64)
            comps = []
            for s, c in enumerate(support):
                if np.abs(dM[s]).max() < 1e-6 * np.abs(dM).max():
                    continue
                sv = np.linalg.svd(dM[s], compute_uv=False)
                rank = int((sv > 1e-5 * sv[0]).sum())
                info = {"component": [c // 6, c % 6], "max_abs_dM": float(np.abs(dM[s]).max()), "max_abs_M": float(np.abs(M[c]).max()),
                        "rank": rank, "positions": int((np.abs(dM[s]).max(axis=1) > 1e-6 * np.abs(dM[s]).max()).sum()),
                        "channels": int((np.abs(dM[s]).max(axis=0) > 1e-6 * np.abs(dM[s]).max()).sum())}
                # 2. wrong B operand of one k tile: column space = V[kt, c] (32 x 4)
                rb = []
                for kt in range(64):
                    q, _ = np.linalg.qr(V[kt, c])
                    rb.append(rel(dM[s] - q @ (q.T @ dM[s]), dM[s]))
                kb = int(np.argmin(rb))
                info["B_operand_fit"] = {"k_tile": kb, "residual": round(rb[kb], 6)}
                if rb[kb] < 1e-3:
                    dU, *_ = np.linalg.lstsq(V[kt if False else kb, c], dM[s], rcond=None)     # [4, 64]
                    Uused = U[kb, c] + dU
                    cands = {"zero": rel(Uused, U[kb, c])}
                    best = (1e9, None)
                    for k2 in range(64):
                        for c2 in ([c] if k2 != kb else []):
                            r = rel(Uused - U[k2, c2], U[kb, c])
                            if r < best[0]:
                                best = (r, [k2, c2 // 6, c2 % 6])
                    for c2 in range(36):
                        for k2 in (kb - 2, kb - 1, kb, kb + 1, kb + 2):
                            if 0 <= k2 < 64 and not (k2 == kb and c2 == c):
                                r = rel(Uused - U[k2, c2], U[kb, c])
                                if r < best[0]:
                                    best = (r, [k2, c2 // 6, c2 % 6])
                    cands["best_other_U_[k_tile, xi, nu]"] = {"which": best[1], "residual": round(best[0], 6)}
                    # per input channel of the k tile: was only one k step (pair half) wrong?
                    info["U_used"] = cands
                    info["dU_rows_max_abs"] = [float(np.abs(dU[i]).max()) for i in range(4)]
                # 3. wrong A operand: row space = U[kt, c] (4 x 64)
                ra = []
                for kt in range(64):
                    q, _ = np.linalg.qr(U[kt, c].T)
                    ra.append(rel(dM[s] - (dM[s] @ q) @ q.T, dM[s]))
                ka = int(np.argmin(ra))
                info["A_operand_fit"] = {"k_tile": ka, "residual": round(ra[ka], 6)}
                if ra[ka] < 1e-3:
                    dV = np.linalg.lstsq(U[ka, c].T, dM[s].T, rcond=None)[0].T                 # [32, 4]
                    Vused = V[ka, c] + dV
                    cands = {"zero": rel(Vused, V[ka, c])}
                    best = (1e9, None)
                    for k2 in range(64):
                        if k2 != ka:
                            r = rel(Vused - V[k2, c], V[ka, c])
                            if r < best[0]:
                                best = (r, k2)
                    cands["best_other_k_tile"] = {"k_tile": best[1], "residual": round(best[0], 6)}
                    info["V_used"] = cands
                # 4. M itself replaced?
                if rb[kb] >= 1e-3 and ra[ka] >= 1e-3:
                    Mbad = M[c] + dM[s]
                    info["M_replaced"] = {"max_abs_M_bad": float(np.abs(Mbad).max()), "rel_to_minus_M": rel(dM[s] + M[c], M[c])}
                comps.append(info)
            rep["components"] = comps
        print(json.dumps(rep), flush=True)
        reports.append(rep)
    return reports




def selftest(path="/tmp/w4_selftest.npz"):
    """The analysis on differences synthesised from known causes (and the restated transforms against a direct convolution)."""
    m = Model()
    b, ty, tx, nt = 1, 3, 5, 2
    V, U, M = m.tile(b, ty, tx, nt)
    ref = m.output(M, nt)
    # direct 3x3 SAME convolution of the tile
    xp = np.pad(m.x[b], ((1, 1), (1, 1), (0, 0)))
    direct = np.zeros((16, 32, 64))
    for ky in range(3):
        for kx in range(3):
            direct += xp[16 * ty + ky:16 * ty + ky + 16, 32 * tx + kx:32 * tx + kx + 32] @ m.w[64 * nt:64 * nt + 64, ky, kx].T
    direct += m.shift[64 * nt:64 * nt + 64]
    print("transforms vs direct convolution: max |diff|", float(np.abs(direct - ref).max()))
    cases = []
    M1 = M.copy(); c, kt = 2 * 6 + 4, 17           # component (2, 4) of k tile 17 multiplied with U of k tile 15
    M1[c] += V[kt, c] @ (U[kt - 2, c] - U[kt, c]); cases.append(M1)
    M2 = M.copy(); kt = 40                           # wave 2's nine components of k tile 40 computed from k tile 38's input
    for c in [(3 + i) * 6 + j for i in range(3) for j in range(3)]:
        M2[c] += (V[kt - 2, c] - V[kt, c]) @ U[kt, c]
    cases.append(M2)
    M3 = M.copy(); M3[7, 8:12] = 0.03                # four positions of component (1, 1) overwritten in the exchange buffer
    cases.append(M3)
    first = ref.astype(np.float32)
    bad = np.array([m.output(Mi, nt) for Mi in cases], dtype=np.float32)
    ids = np.array([[i, b, ty, tx, nt, int((bad[i] != first).sum())] for i in range(len(cases))])
    np.savez_compressed(path, tile_ids=ids, bad=bad, badk=bad, first=np.array([first] * len(cases)),
                        events=json.dumps([]), launches=0, lib="selftest")
    return float(np.abs(direct - ref).max()), analyze(path)


if __name__ == "__main__":
    if sys.argv[1] == "capture":
        capture(int(sys.argv[2]), sys.argv[3])
    elif sys.argv[1] == "selftest":
        selftest()
    else:
        analyze(sys.argv[2])
