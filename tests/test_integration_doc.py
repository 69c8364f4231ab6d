"""INTEGRATION.md quotes ctypes bindings a maintainer of the reference would paste (MeasureVAE/decoder.py:412-453 and the feed
helpers): every `argtypes` list in its fenced python blocks must have the arity of the prototype in include/inpaintnet_hip.h,
and every quoted `lib.inet_*(...)` call of the optimizer sketch too.  (CPU: nothing is called.)"""
import ctypes as C
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_arity():
    """name -> number of parameters of every `inet_*` prototype in the header."""
    hdr = open(os.path.join(REPO, "include", "inpaintnet_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    hdr = re.sub(r"//[^\n]*", " ", hdr)
    out = {}
    for m in re.finditer(r"\b(inet_\w+)\s*\(([^;{}]*?)\)\s*;", hdr, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return out


def python_blocks():
    md = open(os.path.join(REPO, "INTEGRATION.md")).read()
    return re.findall(r"```python\n(.*?)```", md, flags=re.S)


def quoted_argtypes():
    """(name, list) for every `_L.<name>.argtypes = <expr>` of the fenced blocks, the expression evaluated with ctypes."""
    class VaeConfig(C.Structure):
        _fields_ = [("x", C.c_int32)]
    env = {"C": C, "VaeConfig": VaeConfig}
    found = []
    for block in python_blocks():
        text = block.replace("\\\n", " ")
        for m in re.finditer(r"_L\.(inet_\w+)\.argtypes\s*=\s*(.+)", text):
            expr = m.group(2).split("#")[0].strip()
            depth, lines = expr.count("[") - expr.count("]"), [expr]
            rest = text[m.end():].split("\n")[1:]
            while depth > 0 and rest:                       # a list that continues on the next line
                nxt = rest.pop(0).split("#")[0].strip()
                lines.append(nxt)
                depth += nxt.count("[") - nxt.count("]")
            found.append((m.group(1), eval(" ".join(lines), env)))
    return found


def test_header_parser_sees_the_surface():
    ar = header_arity()
    assert ar["inet_abi_version"] == 0
    assert ar["inet_vae_decoder_fwd"] == 15          # cfg, batch, z, target, tf, params, 2 masks, weights, samples, ws, bytes, save, seed, stream
    assert len(ar) >= 57


def test_every_quoted_argtypes_list_has_the_header_arity():
    ar = header_arity()
    quoted = quoted_argtypes()
    names = [n for n, _ in quoted]
    for must in ("inet_vae_decoder_fwd", "inet_vae_decoder_ws_bytes", "inet_vae_param_info", "inet_tokens_to_i64",
                 "inet_split_score"):
        assert must in names, must
    for name, lst in quoted:
        assert name in ar, f"{name} is quoted in INTEGRATION.md but not declared in the header"
        assert len(lst) == ar[name], f"{name}: INTEGRATION.md binds {len(lst)} arguments, the header declares {ar[name]}"


def test_quoted_argtypes_agree_with_the_package_binding():
    """... and, type by type, with the binding the package itself uses (inpaintnet_amd/_lib.py)."""
    from inpaintnet_amd import _lib

    def kind(t):
        if t is None:
            return "p"
        if isinstance(t, type) and issubclass(t, C._Pointer):
            return "p"
        if t in (C.c_void_p, C.c_char_p):
            return "p"
        return "i%d" % C.sizeof(t)
    for name, lst in quoted_argtypes():
        _, mine = _lib._SIGNATURES[name]
        assert [kind(t) for t in lst] == [kind(t) for t in mine], name


def test_quoted_calls_have_the_header_arity():
    ar = header_arity()
    seen = 0
    for block in python_blocks():
        text = block.replace("\\\n", " ")
        for m in re.finditer(r"\b(?:_L|lib)\.(inet_\w+)\(", text):
            depth, i, args, cur = 1, m.end(), [], ""
            while depth:
                ch = text[i]
                depth += ch in "([{"
                depth -= ch in ")]}"
                if depth == 1 and ch == ",":
                    args.append(cur)
                    cur = ""
                elif depth:
                    cur += ch
                i += 1
            if cur.strip():
                args.append(cur)
            assert len(args) == ar[m.group(1)], f"call of {m.group(1)} in INTEGRATION.md passes {len(args)} arguments, header: {ar[m.group(1)]}"
            seen += 1
    assert seen >= 6
