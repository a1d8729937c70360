"""OpenQASM 2 reader for the reference's tensor-network init circuits
(dmrg-to-qc/init_state_circ/*.qasm, the text twin of the QPY files the reference loads
with qiskit at environments/environment_qulacs_TN_notin_agent.py:79-84).  Only what those
files contain: qreg, rx / ry / rz / cx, angle expressions in floats and pi."""
from __future__ import annotations

import ast
import math
import operator
import re

_BIN = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul, ast.Div: operator.truediv}


def _eval(node):
    if isinstance(node, ast.Expression):
        return _eval(node.body)
    if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)):
        return float(node.value)
    if isinstance(node, ast.Name) and node.id == "pi":
        return math.pi
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
        v = _eval(node.operand)
        return -v if isinstance(node.op, ast.USub) else v
    if isinstance(node, ast.BinOp) and type(node.op) in _BIN:
        return _BIN[type(node.op)](_eval(node.left), _eval(node.right))
    raise ValueError("unsupported angle expression")


def parse_angle(expr: str) -> float:
    return _eval(ast.parse(expr.strip(), mode="eval"))


class QasmGate:
    __slots__ = ("name", "qubits", "angle")

    def __init__(self, name, qubits, angle):
        self.name, self.qubits, self.angle = name, tuple(qubits), angle

    def __repr__(self):
        return f"{self.name}({self.angle}) {self.qubits}"


def parse(text: str):
    """-> (n_qubits, [QasmGate]) in file order."""
    n = None
    gates = []
    text = re.sub(r"//[^\n]*", "", text)
    for stmt in text.split(";"):
        s = " ".join(stmt.split())
        if not s or s.startswith("OPENQASM") or s.startswith("include"):
            continue
        m = re.fullmatch(r"qreg (\w+)\[(\d+)\]", s)
        if m:
            if n is not None:
                raise ValueError("more than one qreg")
            n = int(m.group(2))
            continue
        m = re.fullmatch(r"(\w+)\s*(?:\((.*)\))?\s*((?:\w+\[\d+\]\s*,?\s*)+)", s)
        if not m:
            raise ValueError(f"cannot parse statement {s!r}")
        name = m.group(1)
        if name not in ("rx", "ry", "rz", "cx"):
            raise ValueError(f"unsupported gate {name!r}")
        qs = [int(v) for v in re.findall(r"\[(\d+)\]", m.group(3))]
        if n is None or any(q >= n for q in qs):
            raise ValueError("qubit index out of range")
        ang = None if m.group(2) is None else parse_angle(m.group(2))
        if (name == "cx") != (ang is None) or len(qs) != (2 if name == "cx" else 1):
            raise ValueError(f"malformed {name}")
        gates.append(QasmGate(name, qs, ang))
    if n is None:
        raise ValueError("no qreg")
    return n, gates


def layers(n, gates):
    """ASAP layering = qiskit ``circuit_to_dag(c).layers()``; ``len(layers)`` = ``depth()``
    (environment_qulacs_TN_notin_agent.py:102-112)."""
    front = [0] * n
    out = []
    for g in gates:
        d = max(front[q] for q in g.qubits)
        while len(out) <= d:
            out.append([])
        out[d].append(g)
        for q in g.qubits:
            front[q] = d + 1
    return out
