"""Golden fixtures for OpenQASM in/out, made by the real reference: the QASM2 / QASM3 texts it writes for two
circuits, and the circuits (gate class names, final state) it builds from QASM3 programs.

usage: python tests/golden/make_golden_qasm.py
"""

import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import specs  # noqa: E402
from make_golden import import_reference  # noqa: E402


def main():
    dq = import_reference()
    from deepquantum.qasm3 import cir_to_qasm3, qasm3_to_cir

    out = {'export': {}, 'import': {}}
    for name, c in specs.QASM_EXPORT.items():
        cir = specs.build(dq, c['nqubit'], c['spec'])
        if c['measure']:
            cir.measure(wires=c['measure'])
        out['export'][name] = {'qasm2': cir.qasm(), 'qasm3': cir_to_qasm3(cir)}
    programs = dict(specs.QASM3_PROGRAMS)
    programs['roundtrip'] = out['export']['zoo']['qasm3']
    for name, text in programs.items():
        cir = qasm3_to_cir(text)
        state = cir().reshape(-1)
        out['import'][name] = {
            'program': text,
            'gates': [type(op).__name__ for op in cir.operators],
            'wires_measure': list(cir.wires_measure),
            'state': [[float(z.real), float(z.imag)] for z in state],
        }
    with open(os.path.join(HERE, 'golden_qasm.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print('wrote golden_qasm.json;', {k: len(v['gates']) for k, v in out['import'].items()})


if __name__ == '__main__':
    main()
