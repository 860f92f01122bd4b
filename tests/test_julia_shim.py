"""Static check of the Julia shim against the C ABI.

No `julia` binary exists in this image, so `algames.jl_amd/julia/AlgamesHIP.jl` cannot be executed.  What can drift silently is
purely textual: the field lists of its `struct Alg*` mirrors and the `(ret, (argtypes...))` of every `ccall((:alg_x, LIB), ...)`.
This test parses both files and checks names, order, widths and arity against `include/algames_hip.h`
(the reference API the shim keeps: `GameProblem` / `newton_solve!` / `Options`, src/problem/problem.jl:35-53,
src/struct/options.jl:5-116)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "algames_hip.h")
JL = os.path.join(ROOT, "algames.jl_amd", "julia", "AlgamesHIP.jl")

# width classes: (kind, bytes)
C_SCALAR = {"int": ("i", 4), "int32_t": ("i", 4), "int64_t": ("i", 8), "uint64_t": ("i", 8), "double": ("f", 8)}
JL_SCALAR = {"Int32": ("i", 4), "Cint": ("i", 4), "Int64": ("i", 8), "UInt64": ("i", 8), "Float64": ("f", 8), "Cdouble": ("f", 8)}
STRUCT_PAIRS = {"alg_desc": "AlgDesc", "alg_options": "AlgOptions", "alg_record": "AlgRecord", "alg_game_stats": "AlgGameStats"}


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def parse_header():
    text = _strip_comments(open(HDR).read())
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            fm = re.match(r"(\w+)\s+(.*)$", decl, flags=re.S)
            ctype, names = fm.group(1), fm.group(2)
            for nm in names.split(","):
                nm = nm.strip()
                am = re.match(r"(\w+)\s*\[\s*(\w+)\s*\]$", nm)
                if am:
                    fields.append((am.group(1), ctype, am.group(2)))
                else:
                    fields.append((nm, ctype, None))
        structs[m.group(3)] = fields
    macros = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(\w+)\s+(\d+)\b", open(HDR).read())}
    protos = {}
    for m in re.finditer(r"\b(const\s+char\s*\*|int|void)\s+(alg_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        alist = [] if args in ("", "void") else [re.sub(r"\s+", " ", a.strip()) for a in args.split(",")]
        protos[name] = (re.sub(r"\s+", " ", ret), alist)
    return structs, macros, protos


def c_arg_class(arg, structs):
    """('ptr', pointee) or ('i'|'f', bytes) of one C parameter declaration."""
    a = arg.replace("const ", "").strip()
    if "*" in a:
        base = a.split("*")[0].strip()
        depth = a.count("*")
        if base == "alg_handle":
            return ("ptr", "handle" if depth == 1 else "handle*")
        if base == "void":
            return ("ptr", "void")
        if base == "char":
            return ("ptr", "char")
        if base in structs:
            return ("ptr", base)
        return ("ptr", C_SCALAR[base])
    base = a.split()[0]
    return C_SCALAR[base]


def jl_arg_class(t):
    t = t.strip()
    m = re.match(r"(Ptr|Ref)\{(.*)\}$", t)
    if m:
        inner = m.group(2).strip()
        if inner == "Cvoid":
            return ("ptr", "void")
        if inner == "Ptr{Cvoid}":
            return ("ptr", "handle*")
        if inner in STRUCT_PAIRS.values():
            return ("ptr", [c for c, j in STRUCT_PAIRS.items() if j == inner][0])
        return ("ptr", JL_SCALAR[inner])
    if t == "Cstring":
        return ("ptr", "char")
    return JL_SCALAR[t]


def split_top(s):
    """split on commas outside brackets"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out]


def parse_shim():
    text = re.sub(r"#[^\n]*", "", open(JL).read())
    structs = {}
    for m in re.finditer(r"^struct\s+(Alg\w+)\s*\n(.*?)^end", text, flags=re.S | re.M):
        fields = []
        for decl in re.split(r"[;\n]", m.group(2)):
            decl = decl.strip()
            if decl:
                nm, ty = decl.split("::")
                fields.append((nm.strip(), ty.strip()))
        structs[m.group(1)] = fields
    calls = []
    for m in re.finditer(r"ccall\(\(:(\w+),\s*LIB\),\s*", text):
        # return type, then the parenthesised argument-type tuple
        rest = text[m.end():]
        rm = re.match(r"([\w{}]+)\s*,\s*\(", rest)
        assert rm, f"ccall of {m.group(1)}: cannot parse the return type"
        depth, i = 1, rm.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(rest[i], 0); i += 1
        tup = rest[rm.end():i - 1]
        # the actual arguments follow up to the ccall's closing parenthesis
        depth, j = 1, i
        while depth:
            depth += {"(": 1, ")": -1}.get(rest[j], 0); j += 1
        actual = split_top(rest[i:j - 1].lstrip(", \n"))
        calls.append((m.group(1), rm.group(1), [t for t in split_top(tup) if t], actual))
    return structs, calls


def test_struct_mirrors_match_the_header_field_by_field():
    cs, macros, _ = parse_header()
    js, _ = parse_shim()
    for cname, jname in STRUCT_PAIRS.items():
        cf, jf = cs[cname], js[jname]
        assert len(cf) == len(jf), f"{jname}: {len(jf)} fields, {cname} has {len(cf)}"
        for (cn, ct, arr), (jn, jt) in zip(cf, jf):
            assert cn == jn, f"{jname}.{jn} sits where {cname}.{cn} is"
            if arr is not None:
                cnt = macros[arr] if arr in macros else int(arr)
                tm = re.match(r"NTuple\{(\d+),\s*(\w+)\}$", jt)
                assert tm and int(tm.group(1)) == cnt and JL_SCALAR[tm.group(2)] == C_SCALAR[ct], f"{jname}.{jn}: {jt} vs {ct}[{cnt}]"
            elif ct in cs:
                assert STRUCT_PAIRS.get(ct) == jt, f"{jname}.{jn}: {jt} vs {ct}"
            else:
                assert JL_SCALAR[jt] == C_SCALAR[ct], f"{jname}.{jn}: {jt} vs {ct}"
    # every struct the shim mirrors is a header struct; alg_step_info (step-wise inspection entry) is not used by the shim
    assert set(js) == set(STRUCT_PAIRS.values())


def test_every_ccall_matches_its_prototype():
    cs, _, protos = parse_header()
    _, calls = parse_shim()
    assert len(calls) >= 40
    for name, ret, argt, actual in calls:
        assert name in protos, f"ccall of {name}: not declared in include/algames_hip.h"
        cret, cargs = protos[name]
        if cret == "int":
            assert JL_SCALAR[ret] == ("i", 4), f"{name}: returns {ret}"
        elif cret == "void":
            assert ret == "Cvoid", f"{name}: returns {ret}"
        else:
            assert ret == "Cstring", f"{name}: returns {ret}"
        assert len(argt) == len(cargs), f"{name}: {len(argt)} argument types for {len(cargs)} parameters ({cargs})"
        assert len(actual) == len(argt), f"{name}: {len(actual)} arguments passed for {len(argt)} declared types"
        for k, (jt, ca) in enumerate(zip(argt, cargs)):
            jc, cc = jl_arg_class(jt), c_arg_class(ca, cs)
            if cc == ("ptr", "handle"):
                assert jc == ("ptr", "void"), f"{name} arg {k}: {jt} for {ca}"
            else:
                assert jc == cc, f"{name} arg {k}: {jt} ({jc}) for `{ca}` ({cc})"


def test_the_shim_binds_the_entry_points_of_the_hot_path():
    _, calls = parse_shim()
    used = {c[0] for c in calls}
    for need in ("alg_create", "alg_destroy", "alg_set_options", "alg_set_x0", "alg_set_lqr", "alg_newton_solve", "alg_newton_solve_async",
                 "alg_get_traj", "alg_set_traj", "alg_get_con_duals", "alg_get_history", "alg_get_stats", "alg_ibr_newton_solve",
                 "alg_mpc_solve", "alg_last_error"):
        assert need in used, need
