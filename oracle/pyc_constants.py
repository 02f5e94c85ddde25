"""ORACLE TOOLING (test infrastructure, NOT product code): pins what can be pinned of the reference's bytecode-only callers WITHOUT
running them.  `evaluator.py` and `datasets/mask_generator.py` exist in /root/reference only as CPython 3.9 / 3.12 bytecode
(`__pycache__/evaluator.cpython-39.pyc`, `datasets/__pycache__/mask_generator.cpython-39.pyc`); the image's interpreter is 3.10, so
they cannot be imported.  This script reads the 3.9 files AS DATA - `marshal.loads` of the bytes behind the 16-byte header yields code
objects, nothing is ever exec'ed, imported or called - walks the nested code objects and writes, per function, its constants
(`co_consts` without the nested code), names (`co_names`), argument names and first line to
`tests/golden/evaluator_constants.json`.  tests/test_pyc_constants_cpu.py then asserts the product's header string, result keys,
interpolation keyword sets, thresholds, defaults and CLI choices against that file (the .pyc themselves do not travel).

    python oracle/pyc_constants.py            (needs /root/reference; the committed JSON is its output)
"""
import json
import marshal
import os
import sys
import types

REF = "/root/reference"
SOURCES = {"evaluator": "__pycache__/evaluator.cpython-39.pyc",
           "mask_generator": "datasets/__pycache__/mask_generator.cpython-39.pyc"}
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "evaluator_constants.json")


def plain(c):
    """a constant as JSON: tuples / frozensets become lists, Ellipsis '...', bytes hex; code objects are dropped by the caller"""
    if c is Ellipsis:
        return "..."
    if isinstance(c, (tuple, frozenset)):
        return [plain(x) for x in (sorted(c, key=repr) if isinstance(c, frozenset) else c)]
    if isinstance(c, bytes):
        return {"bytes": c.hex()}
    if isinstance(c, complex):
        return {"complex": [c.real, c.imag]}
    return c


def walk(co, prefix=""):
    name = prefix + co.co_name
    yield name, co
    for c in co.co_consts:
        if isinstance(c, types.CodeType):
            yield from walk(c, name + ".")


def read_code(path):
    with open(path, "rb") as f:
        raw = f.read()
    return raw[:4].hex(), marshal.loads(raw[16:])  # magic, flags, mtime, size: 16 bytes since 3.7


def main():
    out = {"_how": "marshal.loads of the 3.9 .pyc bodies read as data (oracle/pyc_constants.py); nothing executed", "_python": sys.version.split()[0]}
    for mod, rel in SOURCES.items():
        magic, code = read_code(os.path.join(REF, rel))
        funcs = {}
        for name, co in walk(code):
            funcs[name] = {"firstlineno": co.co_firstlineno, "argcount": co.co_argcount, "kwonlyargcount": co.co_kwonlyargcount,
                           "argnames": list(co.co_varnames[:co.co_argcount + co.co_kwonlyargcount]),
                           "consts": [plain(c) for c in co.co_consts if not isinstance(c, types.CodeType)],
                           "names": list(co.co_names)}
        out[mod] = {"source": rel, "magic": magic, "functions": funcs}
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print(OUT, {m: len(out[m]["functions"]) for m in SOURCES})


if __name__ == "__main__":
    main()
