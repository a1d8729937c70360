"""CPU model of the layout scheduler (vqe_reg.h schedule_ops): counts LDS re-layouts per circuit
for the in-order greedy scheduler and for a commutation-aware list scheduler."""
import sys, numpy as np

def par(v): return bin(v).count("1") & 1

def compile_ops(n, kind, q0, q1):
    rowA = [1 << i for i in range(n)]      # row q of A
    colI = [1 << i for i in range(n)]      # column q of A^-1  (xm for logical q)
    ops = []
    for k, a, b in zip(kind, q0, q1):
        if k == 0:      # CNOT a->b : row_b ^= row_a ; col_a(inv) ^= col_b(inv)
            rowA[b] ^= rowA[a]; colI[a] ^= colI[b]
        elif k in (1, 2, 3):
            xm = colI[a] if k != 3 else 0
            zm = rowA[a] if k != 1 else 0
            ops.append((xm, zm))
    return ops

class Basis:
    def __init__(s): s.e = []
    def reduce(s, v):
        for e in s.e:
            if v & (e & -e) and False: pass
        # simple gaussian: keep echelon by highest bit
        for e in s.e:
            hb = 1 << (e.bit_length() - 1)
            if v & hb: v ^= e
        return v
    def add(s, v):
        v = s.reduce(v)
        if v:
            s.e.append(v); s.e.sort(reverse=True)
            # keep reduced: not needed for span test as long as we reduce in descending order
            return True
        return False

def greedy(ops, R, n):
    nl = 1
    def open_layout(frm):
        B = Basis()
        for x, z in ops[frm:]:
            if len(B.e) >= R: break
            if x: B.add(x)
        return B
    B = open_layout(0)
    for o, (x, z) in enumerate(ops):
        if x and B.reduce(x):
            B = open_layout(o); nl += 1
    return nl

def anti(a, b): return par(a[0] & b[1]) ^ par(a[1] & b[0])

def listsched(ops, R, n, lookahead=True):
    pend = list(range(len(ops)))
    nl = 0
    B = None
    while pend:
        # ready ops: commute with all earlier pending ops
        progressed = True
        while progressed and B is not None:
            progressed = False
            newp = []
            blockers = []
            for k in pend:
                ready = all(not anti(ops[k], ops[j]) for j in blockers)
                x = ops[k][0]
                if ready and (x == 0 or B.reduce(x) == 0):
                    progressed = True      # executed
                else:
                    blockers.append(k); newp.append(k)
            pend = newp
        if not pend: break
        # open a new layout: take directions from pending ops in order, preferring ready ones
        B = Basis(); nl += 1
        blockers = []
        cand_ready, cand_rest = [], []
        for k in pend:
            ready = all(not anti(ops[k], ops[j]) for j in blockers)
            (cand_ready if ready else cand_rest).append(k)
            blockers.append(k)
        for k in (cand_ready + cand_rest) if lookahead else pend:
            if len(B.e) >= R: break
            if ops[k][0]: B.add(ops[k][0])
        if not B.e:     # only diagonal ops left
            pass
    return max(nl, 1)

if __name__ == "__main__":
    n, G, Bn = 12, 64, 300
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    rng = np.random.default_rng(1)
    g1 = g2 = g3 = 0; nrot = 0
    for b in range(Bn):
        is_cnot = rng.random(G) < 0.5
        c = rng.integers(0, n, G); t = (c + 1 + rng.integers(0, n - 1, G)) % n
        rq = rng.integers(0, n, G); rk = rng.integers(1, 4, G)
        kind = np.where(is_cnot, 0, rk); q0 = np.where(is_cnot, c, rq); q1 = np.where(is_cnot, t, -1)
        ops = compile_ops(n, kind, q0, q1)
        nrot += len(ops)
        g1 += greedy(ops, R, n); g2 += listsched(ops, R, n, False); g3 += listsched(ops, R, n, True)
    print(f"R={R} rot/circ {nrot/Bn:.1f}  layouts: greedy {g1/Bn:.2f}  list {g2/Bn:.2f}  list+ready-first {g3/Bn:.2f}")
