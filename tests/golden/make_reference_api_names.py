"""Names of the reference's API surface that julia/MadNLPHIP.jl binds to -- generated HERE from /root/reference (it does not travel to the GPU
box); the fixture holds identifiers only: function / type / constant names defined anywhere under src/, and the field names of the structs
the glue reads fields of.  tests/test_julia_glue.py checks every `MadNLP.<name>`, every imported name and every field access of the glue
against it (the glue cannot be executed: no Julia toolchain in any image).
usage: python tests/golden/make_reference_api_names.py  ->  tests/golden/reference_api_names.json"""
import json
import os
import re

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_api_names.json")
STRUCTS = ["SchurComplementKKTSystem", "SparseCondensedKKTSystem", "DenseCondensedKKTSystem", "SparseKKTSystem", "SparseUnreducedKKTSystem",
           "DenseKKTSystem", "SparseMatrixCOO", "SparseCallback", "DenseCallback", "MadNLPSolver", "MadNLPOptions", "LapackCPUSolver",
           "UnreducedKKTVector"]

names, fields, arities = set(), {}, {}


def positional_arity(src, open_idx):
    """Number of positional parameters of the signature whose '(' is at src[open_idx] (top-level commas before ';'); None if unbalanced."""
    depth, n, seen, i = 0, 0, False, open_idx
    while i < len(src):
        c = src[i]
        if c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
            if depth == 0:
                return n + (1 if seen else 0)
        elif depth == 1:
            if c == ",":
                n += 1
                seen = False
            elif c == ";":
                return n + (1 if seen else 0)
            elif not c.isspace():
                seen = True
        i += 1
    return None

for root, _dirs, files in os.walk(os.path.join(REF, "src")):
    for f in files:
        if not f.endswith(".jl"):
            continue
        text = open(os.path.join(root, f), encoding="utf-8").read()
        text = re.sub(r"#=.*?=#", "", text, flags=re.S)
        lines = [re.sub(r"#.*$", "", ln) for ln in text.splitlines()]
        src = "\n".join(lines)
        for m in re.finditer(r"\bfunction\s+(?:[A-Za-z_][A-Za-z_0-9]*\.)*([A-Za-z_][A-Za-z_0-9!]*)\s*[\({]", src):
            names.add(m.group(1))
        # positional arities of the methods (long and short form)
        for m in re.finditer(r"(?:\bfunction\s+|^\s*(?:@inline\s+)?)(?:[A-Za-z_][A-Za-z_0-9]*\.)*([A-Za-z_][A-Za-z_0-9!]*)(?:\{[^}\n]*\})?\(", src, flags=re.M):
            a = positional_arity(src, m.end() - 1)
            if a is not None:
                arities.setdefault(m.group(1), set()).add(a)
        for m in re.finditer(r"^\s*(?:@inline\s+)?(?:[A-Za-z_][A-Za-z_0-9]*\.)*([A-Za-z_][A-Za-z_0-9!]*)\((?:[^()]|\([^()]*\))*\)(?:\s+where\s+[^=\n]+)?\s*=(?!=)", src, flags=re.M):
            names.add(m.group(1))
        for m in re.finditer(r"\b(?:abstract\s+type|primitive\s+type|mutable\s+struct|struct)\s+([A-Za-z_][A-Za-z_0-9]*)", src):
            names.add(m.group(1))
        for m in re.finditer(r"^\s*(?:const\s+)?([A-Za-z_][A-Za-z_0-9]*)\s*=\s*", src, flags=re.M):
            names.add(m.group(1))
        for m in re.finditer(r"@enum\s+([A-Za-z_][A-Za-z_0-9]*)\s*(?:::\s*\w+\s*)?begin(.*?)end", src, flags=re.S):   # enums and their values
            names.add(m.group(1))
            for v in re.finditer(r"^\s*([A-Za-z_][A-Za-z_0-9]*)", m.group(2), flags=re.M):
                names.add(v.group(1))
        for m in re.finditer(r"@enum\s*\(?\s*([A-Za-z_][A-Za-z_0-9]*)(?:::\w+)?\s*,?([^\n\)]*)", src):
            names.add(m.group(1))
            for v in re.finditer(r"([A-Za-z_][A-Za-z_0-9]*)\s*(?:=\s*\d+)?", m.group(2)):
                names.add(v.group(1))
        for m in re.finditer(r"\bmodule\s+([A-Za-z_][A-Za-z_0-9]*)", src):
            names.add(m.group(1))
        # names brought into the module's namespace (`import NLPModels: AbstractNLPModel, ...`: MadNLP.AbstractNLPModel resolves)
        for m in re.finditer(r"^\s*(?:import|using)\s+[A-Za-z_][A-Za-z_0-9.]*\s*:\s*([^\n]+(?:,\s*\n[^\n]+)*)", src, flags=re.M):
            for v in re.finditer(r"([A-Za-z_][A-Za-z_0-9!]*)", m.group(1)):
                names.add(v.group(1))
        for st in STRUCTS:
            m = re.search(r"(?:mutable\s+)?struct\s+" + st + r"\b[^\n]*\n(.*?)\n\s*end\b", src, flags=re.S)
            if m:
                fl = []
                for ln in m.group(1).splitlines():
                    mm = re.match(r"^\s*([a-zA-Z_][A-Za-z_0-9]*)\s*(?:::|$)", ln)
                    if mm and mm.group(1) not in ("function", "end", "new", "return"):
                        fl.append(mm.group(1))
                fields[st] = fl
json.dump({"source": "identifiers of /root/reference/src/**/*.jl (MadNLP.jl): defined function / type / constant / enum names, and the field names of the structs "
                     "julia/MadNLPHIP.jl reads -- names only, written by tests/golden/make_reference_api_names.py",
           "names": sorted(names), "struct_fields": fields,
           "method_arities": {k: sorted(v) for k, v in sorted(arities.items()) if k in names}}, open(OUT, "w"), indent=0)
print(len(names), "names;", {k: len(v) for k, v in fields.items()})
