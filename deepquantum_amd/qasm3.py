"""OpenQASM in and out: the wire formats on either side of the statevector path.

* :func:`cir_to_qasm2` -- what ``QubitCircuit.qasm()`` returns (reference: circuit.py:570-627 and the per-gate
  ``_qasm`` strings of gate.py): OpenQASM 2.0 over ``qelib1.inc``, with the three composite definitions the
  reference emits on first use (``cs``, ``csdg``, ``ryy``).
* :func:`cir_to_qasm3` -- OpenQASM 3.0 with ``ctrl @`` modifiers (reference: qasm3.py:40-151).
* :func:`qasm3_to_cir` -- an interpreter for the subset the reference reads (qasm3.py:166-471): the standard
  gates, ``def`` blocks used as gate definitions (parameters and qubit arguments), the modifiers ``inv @``,
  ``ctrl @`` and ``pow(k) @`` (integer powers repeat, fractional powers go through an eigendecomposition),
  ``barrier`` and ``measure``.

The emitters are table driven and the reader is a small recursive interpreter with its own arithmetic evaluator
(no ``eval``); the texts and the circuits they produce are checked against the reference's
(tests/golden/golden_qasm.json).  Deliberate differences, all places where the reference contradicts OpenQASM:
its reader ignores ``inv @`` (a double sign flip cancels it: ``inv @ rx(0.3)`` applies ``rx(+0.3)``, and a negative
power of a ``def`` gate reverses the body without inverting its gates), it drops one-line ``def`` blocks, and its
QASM3 writer negates the three angles of an inverted ``u`` without exchanging phi and lambda.  Here inversion means
inversion (tests check ``G inv@G = 1``).
"""

from __future__ import annotations

import ast
import math
import operator
import re
from typing import Any

import torch

from .circuit import QubitCircuit
from .gate import (
    CNOT, Barrier, Fredkin, Hadamard, PauliX, PauliY, PauliZ, PhaseShift, Rx, Rxx, Ry, Ryy, Rz, Rzz, SDaggerGate,
    SGate, Swap, TDaggerGate, TGate, Toffoli, U3Gate,
)
from .operation import Channel, Gate, Layer, Operation

# name, whether a singly-controlled form exists in qelib1 (or is defined on first use)
_QASM2_NAME = {U3Gate: 'u', PhaseShift: 'p', PauliX: 'x', PauliY: 'y', PauliZ: 'z', Hadamard: 'h', SGate: 's',
               SDaggerGate: 'sdg', TGate: 't', TDaggerGate: 'tdg', Rx: 'rx', Ry: 'ry', Rz: 'rz', Swap: 'swap',
               Rxx: 'rxx', Ryy: 'ryy', Rzz: 'rzz', CNOT: 'cx', Toffoli: 'ccx', Fredkin: 'cswap'}
_NO_CONTROLS = (TGate, TDaggerGate, CNOT, Rxx, Ryy, Rzz, Toffoli, Fredkin, Barrier)
_ONE_CONTROL = (U3Gate, PhaseShift, PauliY, PauliZ, Hadamard, SGate, SDaggerGate, Rx, Ry, Rz, Swap)
_DEFINITIONS = {
    'cs': 'gate cs q0,q1 { p(pi/4) q0; cx q0,q1; p(-pi/4) q1; cx q0,q1; p(pi/4) q1; }\n',
    'csdg': 'gate csdg q0,q1 { p(-pi/4) q0; cx q0,q1; p(pi/4) q1; cx q0,q1; p(-pi/4) q1; }\n',
    'ryy': ('gate ryy(param0) q0,q1 { rx(pi/2) q0; rx(pi/2) q1; cx q0,q1; rz(param0) q1; cx q0,q1; '
            'rx(-pi/2) q0; rx(-pi/2) q1; }\n'),
}


def _angles(op: Gate) -> list[float]:
    """The gate's parameters as Python floats, negated / reordered for an inverted gate."""
    if isinstance(op, U3Gate):
        vals = [op.theta.item(), op.phi.item(), op.lambd.item()]
        return [-vals[0], -vals[2], -vals[1]] if op.inv_mode else vals
    if getattr(op, 'npara', 0) > 0 and hasattr(op, 'theta'):
        val = op.theta.item()
        return [-val if op.inv_mode else val]
    return []


def _qasm2_line(op: Gate, defined: set[str]) -> str:
    if isinstance(op, Barrier):
        return 'barrier ' + ','.join(f'q[{w}]' for w in op.wires) + ';\n'
    name = _QASM2_NAME[type(op)]
    qubits = [f'q[{w}]' for w in op.controls + op.wires]
    params = _angles(op)
    if op.controls:
        if isinstance(op, PauliX):
            name = {1: 'cx', 2: 'ccx'}.get(len(op.controls), f'c{len(op.controls)}x')
        else:
            name = 'c' + name
        if isinstance(op, U3Gate):
            params = params + [0.0]                       # cu(theta, phi, lambda, gamma)
    text = ''
    if name in _DEFINITIONS and name not in defined:
        defined.add(name)
        text = _DEFINITIONS[name]
    arg = '(' + ','.join(str(p) for p in params) + ')' if params else ''
    return text + f'{name}{arg} ' + ','.join(qubits) + ';\n'


def cir_to_qasm2(circuit: QubitCircuit) -> str:
    """OpenQASM 2.0 text of the circuit; raises ``ValueError`` for what qelib1 cannot express, exactly where the
    reference does (circuit.py:596-621)."""
    out = ['OPENQASM 2.0;\ninclude "qelib1.inc";\n', f'qreg q[{circuit.nqubit}];\n']
    if circuit.wires_measure or circuit.wires_condition:
        out.append(f'creg c[{circuit.nqubit}];\n')
    defined: set[str] = set()
    for op in circuit.operators:
        if not isinstance(op, tuple(_QASM2_NAME) + (Barrier,)):
            raise ValueError(f'{op.name} is NOT supported')
        if op.condition:
            raise ValueError(f'Conditional mode is NOT supported for {op.name}')
        limit = 4 if isinstance(op, PauliX) else 0 if isinstance(op, _NO_CONTROLS) else 1
        if len(op.controls) > limit:
            raise ValueError(f'Too many control bits for {op.name}')
        out.append(_qasm2_line(op, defined))
    out += [f'measure q[{w}] -> c[{w}];\n' for w in circuit.wires_measure]
    return ''.join(out)


def _qasm3_line(op: Operation) -> str:
    if isinstance(op, Layer):
        return '\n'.join(_qasm3_line(g) for g in op.gates)
    if isinstance(op, Barrier):
        return 'barrier ' + ', '.join(f'q[{w}]' for w in op.wires) + ';'
    if isinstance(op, Channel):
        return f'// Quantum channels like {op.name} are not part of the OpenQASM 3.0 core specification.'
    if not isinstance(op, Gate):
        return f'// Unsupported operation type: {op.__class__.__name__}'
    name = _QASM2_NAME.get(type(op))
    if name is None:
        return f'// Unsupported gate: {op.name}'
    params = _angles(op)     # (an inverted U3 is u(-theta, -lambda, -phi); the reference's QASM3 writer forgets the swap)
    arg = '(' + ', '.join(str(p) for p in params) + ')' if params else ''
    if isinstance(op, (CNOT, Toffoli, Fredkin)):
        return f'{name} ' + ', '.join(f'q[{w}]' for w in op.wires) + ';'
    qubits = ', '.join(f'q[{w}]' for w in op.controls + op.wires)
    return 'ctrl @ ' * len(op.controls) + f'{name}{arg} {qubits};'


def cir_to_qasm3(circuit: QubitCircuit) -> str:
    """OpenQASM 3.0 text of the circuit (reference: qasm3.py:117-151)."""
    out = ['OPENQASM 3.0;', 'include "stdgates.inc";', f'qubit[{circuit.nqubit}] q;']
    if circuit.wires_measure:
        out.append(f'bit[{max(circuit.wires_measure) + 1}] c;')
    out += [line for line in (_qasm3_line(op) for op in circuit.operators) if line]
    if circuit.wires_measure:
        out.append('\n// Measurements')
        out += [f'c[{w}] = measure q[{w}];' for w in sorted(circuit.wires_measure)]
    return '\n'.join(out)


# ---------------------------------------------------------------------------------------------------------
# reader
# ---------------------------------------------------------------------------------------------------------
_BINOPS = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul, ast.Div: operator.truediv,
           ast.Pow: operator.pow, ast.Mod: operator.mod, ast.FloorDiv: operator.floordiv}


def _evaluate(expr: str, scope: dict[str, float]) -> float:
    """Arithmetic on numbers, ``pi`` and the formal parameters in ``scope``."""
    def walk(node):
        if isinstance(node, ast.Expression):
            return walk(node.body)
        if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)):
            return node.value
        if isinstance(node, ast.Name):
            if node.id == 'pi':
                return math.pi
            return scope[node.id]
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
            val = walk(node.operand)
            return -val if isinstance(node.op, ast.USub) else val
        if isinstance(node, ast.BinOp) and type(node.op) in _BINOPS:
            return _BINOPS[type(node.op)](walk(node.left), walk(node.right))
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id == 'np' \
                and node.attr == 'pi':
            return math.pi
        raise ValueError(f'unsupported expression {expr!r}')

    return float(walk(ast.parse(expr.strip(), mode='eval')))


class _Definition:
    def __init__(self, params: list[str], qubits: list[str], body: list[str]) -> None:
        self.params, self.qubits, self.body = params, qubits, body


_CALL = re.compile(r'((?:(?:inv|ctrl|pow\s*\(.*?\))\s*@\s*)*)(\w+)(?:\((.*?)\))?\s+(.*?);')
_INVERSE_NAME = {'s': 'sdg', 'sdg': 's', 't': 'tdg', 'tdg': 't'}
_ROTATIONS = ('rx', 'ry', 'rz', 'p', 'rxx', 'ryy', 'rzz')


def _split_definitions(lines: list[str]) -> tuple[dict[str, _Definition], list[str]]:
    """Pull the ``def name(params) qubits { ... }`` blocks out of the program."""
    defs: dict[str, _Definition] = {}
    main: list[str] = []
    i = 0
    while i < len(lines):
        line = lines[i]
        if not line.startswith('def '):
            main.append(line)
            i += 1
            continue
        header = line
        if '{' not in header:
            if i + 1 < len(lines) and lines[i + 1] == '{':
                i += 1
            else:                                  # not a block: leave it to the statement loop
                main.append(header)
                i += 1
                continue
        opened = '{' in header
        depth, body = (header.count('{') - header.count('}')) if opened else 1, []
        i += 1
        while i < len(lines) and depth > 0:
            depth += lines[i].count('{') - lines[i].count('}')
            if depth > 0:
                body.append(lines[i])
            i += 1
        head = header[3:].split('{')[0].strip()
        m = re.match(r'(\w+)\s*\((.*?)\)\s*(.*)', head)
        if m:
            name, params, qubits = m.groups()
        else:
            name, qubits = re.match(r'(\w+)\s*(.*)', head).groups()
            params = ''
        # statements written on the header line after the brace, or several per line, are split on ';'
        inline = header.split('{', 1)[1] if '{' in header else ''
        stmts = [s_.strip() + ';' for chunk in [inline] + body for s_ in chunk.replace('}', '').split(';') if s_.strip()]
        defs[name] = _Definition([p.strip() for p in params.split(',') if p.strip()],
                                 [q.strip() for q in qubits.split(',') if q.strip()], stmts)
    return defs, main


def qasm3_to_cir(qasm_string: str) -> QubitCircuit:
    """Build a ``QubitCircuit`` from OpenQASM 3.0 text (reference: qasm3.py:166-471)."""
    lines = [ln.split('//')[0].strip() for ln in qasm_string.strip().splitlines()]
    lines = [ln for ln in lines if ln]
    if not any(ln.startswith('OPENQASM 3.0') for ln in lines):
        raise ValueError('Input is not a valid OpenQASM 3.0 string (Header missing).')
    defs, main = _split_definitions(lines)
    nqubit = 0
    for ln in main:
        m = re.search(r'qubit\[(\d+)\]', ln)
        if m:
            nqubit = int(m.group(1))
            break
    if nqubit == 0:
        raise ValueError('Qubit declaration not found or zero qubits specified.')
    cir = QubitCircuit(nqubit=nqubit)

    def index(q: str) -> int:
        return int(q[q.index('[') + 1: q.index(']')])

    def unitary_of(name: str, params: str, nq: int) -> torch.Tensor:
        """Matrix of one gate call on ``nq`` fresh qubits (for fractional powers)."""
        prog = ['OPENQASM 3.0;', f'qubit[{nq}] q;']
        for dname, d in defs.items():
            plist = f'({",".join(d.params)})' if d.params else ''
            prog.append(f'def {dname}{plist} {",".join(d.qubits)} {{')
            prog += d.body
            prog.append('}')
        prog.append(f'{name}{"(" + params + ")" if params else ""} ' + ', '.join(f'q[{i}]' for i in range(nq)) + ';')
        sub = qasm3_to_cir('\n'.join(prog))
        # the matrix comes from the gate kernels (QubitCircuit.get_unitary): on the GPU, like everything else
        from . import backend
        if backend.get_test_backend() is None and torch.cuda.is_available():
            sub = sub.to('cuda')
        return sub.get_unitary().cpu()

    def builtin(name: str, values: list[float], qubits: list[int], nctrl: int, outer: list[int], inverted: bool) -> None:
        controls, rest = outer + qubits[:nctrl], qubits[nctrl:]
        if inverted:
            if name in _ROTATIONS:
                values = [-v for v in values]
            elif name == 'u':
                values = [-values[0], -values[2], -values[1]]
            name = _INVERSE_NAME.get(name, name)
        if name in ('cx', 'cz'):
            controls, target = controls + [rest[0]], rest[1]
            if name == 'cx' and len(controls) == 1:
                cir.cnot(controls[0], target)
            else:
                (cir.x if name == 'cx' else cir.z)(target, controls=controls)
        elif name == 'ccx':
            controls, target = controls + rest[:2], rest[2]
            if len(controls) == 2:
                cir.toffoli(controls[0], controls[1], target)
            else:
                cir.x(target, controls=controls)
        elif name == 'cswap':
            controls, pair = controls + [rest[0]], rest[1:3]
            if len(controls) == 1:
                cir.fredkin(controls[0], pair[0], pair[1])
            else:
                cir.swap(pair, controls=controls)
        else:
            wires: Any = rest[0] if len(rest) == 1 else rest
            if name in ('h', 'x', 'y', 'z', 's', 'sdg', 't', 'tdg', 'swap'):
                getattr(cir, name)(wires, controls=controls)
            elif name in _ROTATIONS:
                getattr(cir, name)(wires, inputs=values, controls=controls)
            elif name == 'u':
                cir.u3(wires, inputs=values, controls=controls)
            else:
                print(f"Warning: Unsupported built-in gate '{name}'")

    def run(stmts: list[str], scope: dict[str, float], outer: list[int], inverted: bool) -> None:
        for stmt in (reversed(stmts) if inverted else stmts):
            stmt = stmt.strip()
            if not stmt or stmt.startswith(('OPENQASM', 'include', 'qubit', 'bit', 'defcal')):
                continue
            if 'measure' in stmt:
                for w in re.findall(r'q\[(\d+)\]', stmt):
                    if int(w) not in cir.wires_measure:
                        cir.wires_measure.append(int(w))
                continue
            if stmt.startswith('barrier'):
                args = stmt[len('barrier'):].replace(';', '').strip()
                cir.barrier(wires=[index(q) for q in args.split(',')] if args else None)
                continue
            m = _CALL.match(stmt)
            if not m:
                print(f"Warning: Could not parse line: '{stmt}'")
                continue
            mods, name, params, qubit_text = m.groups()
            params = params or ''
            nctrl = mods.count('ctrl')
            power = 1.0
            pm = re.search(r'pow\s*\((.*?)\)', mods)
            if pm:
                power = _evaluate(pm.group(1), {})
            # G^p in an inverted context / under `inv @` / with p < 0: each of the three flips the direction once
            flip = inverted ^ (mods.count('inv') % 2 == 1) ^ (power < 0)
            power = abs(power)
            qtext = [q.strip() for q in qubit_text.split(',')]
            if int(power) != power:
                # fractional power: U^p = V diag(lambda^p) V^-1 on the gate's own qubits
                base = unitary_of(name, params, len(qtext) - nctrl).to(torch.cfloat)
                lam, vec = torch.linalg.eig(base)
                mat = vec @ torch.diag(lam ** (-power if flip else power)) @ torch.linalg.inv(vec)
                cir.any(mat, wires=[index(q) for q in qtext[nctrl:]], controls=outer + [index(q) for q in qtext[:nctrl]])
                continue
            for _ in range(int(power)):
                if name in defs:
                    d = defs[name]
                    if len(qtext) - nctrl != len(d.qubits):
                        print(f"Warning: Mismatched qubit count for gate '{name}'.")
                        continue
                    values = [_evaluate(p, scope) for p in params.split(',')] if params else []
                    if len(values) != len(d.params):
                        print(f"Warning: Mismatched parameter count for gate '{name}'.")
                        continue
                    inner = dict(scope)
                    inner.update(zip(d.params, values, strict=True))
                    rename = dict(zip(d.qubits, qtext[nctrl:], strict=True))
                    body = []
                    for b in d.body:
                        for formal, value in inner.items():
                            b = re.sub(r'\b' + re.escape(formal) + r'\b', str(value), b)
                        for formal, actual in rename.items():
                            b = re.sub(r'\b' + re.escape(formal) + r'\b', actual, b)
                        body.append(b)
                    run(body, inner, outer + [index(q) for q in qtext[:nctrl]], flip)
                else:
                    values = [_evaluate(p, scope) for p in params.split(',')] if params else []
                    builtin(name, values, [index(q) for q in qtext], nctrl, outer, flip)

    run(main, {}, [], False)
    cir.wires_measure.sort()
    return cir
