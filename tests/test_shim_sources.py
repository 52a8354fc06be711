"""The Rust shim cannot be compiled here (no Rust toolchain): what CAN be checked is that its sources do not drift from the C ABI.
Every `extern "C"` declaration in integration/rust_shim/src/lib.rs and in INTEGRATION.md section 2 is parsed and compared with
include/fheaes.h: name, argument count, and for every argument and the return value whether it is a pointer (and to what: ctx,
params, u64, u32, char, a pointer to a pointer) or a scalar (and its width).  A host-path function of the header that the shim
does not bind is a failure too."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

# the functions a drop-in for src/server/sbox + src/server/server.rs needs (lifetime, keys, plugin API, Server API)
HOST_PATH = {"fheaes_create", "fheaes_destroy", "fheaes_last_error", "fheaes_upload_keys", "fheaes_clone_keys", "fheaes_synchronize",
             "fheaes_wopbs_batch", "fheaes_sbox", "fheaes_many_sbox", "fheaes_aes_key_expansion", "fheaes_aes_encrypt",
             "fheaes_aes_decrypt", "fheaes_add_scalar"}


def _kind_c(t: str) -> str:
    t = re.sub(r"\bconst\b", "", t).strip()
    stars = t.count("*")
    base = t.replace("*", "").strip()
    base = {"fheaes_ctx": "ctx", "fheaes_params": "params", "uint64_t": "u64", "uint32_t": "u32", "char": "char", "int": "i32", "double": "f64",
            "void": "void", "size_t": "usize"}.get(base, base)
    return "*" * stars + base


def _kind_rs(t: str) -> str:
    t = t.strip()
    stars = 0
    while True:
        m = re.match(r"\*(?:const|mut)\s+(.*)", t)
        if not m:
            break
        stars += 1
        t = m.group(1).strip()
    base = {"fheaes_ctx": "ctx", "FheaesCtx": "ctx", "fheaes_params": "params", "FheaesParams": "params", "u64": "u64", "u32": "u32", "c_char": "char",
            "c_int": "i32", "i32": "i32", "f64": "f64", "c_void": "void", "usize": "usize"}.get(t, t)
    return "*" * stars + base


def header_decls():
    text = (ROOT / "include" / "fheaes.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(fheaes_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        kinds = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                typ = re.sub(r"\b[A-Za-z_][A-Za-z0-9_]*$", "", a).strip() if not a.endswith("*") else a      # drop the parameter name
                kinds.append(_kind_c(typ))
        out[name] = (_kind_c(ret), kinds)
    return out


def rust_decls(text: str):
    out = {}
    for block in re.findall(r'extern\s+"C"\s*\{(.*?)\n\}', text, flags=re.S):
        for m in re.finditer(r"pub\s+fn\s+(fheaes_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
            name, args, ret = m.group(1), m.group(2), (m.group(3) or "void").strip()
            kinds = [_kind_rs(a.split(":", 1)[1]) for a in args.split(",") if ":" in a]
            out[name] = ("void" if ret == "void" else _kind_rs(ret), kinds)
    return out


def _check(src_name: str, decls: dict, header: dict):
    assert decls, "no extern \"C\" declarations found in %s" % src_name
    for name, (ret, kinds) in decls.items():
        assert name in header, "%s binds %s, which include/fheaes.h does not declare" % (src_name, name)
        hret, hkinds = header[name]
        assert len(kinds) == len(hkinds), "%s: %s takes %d arguments, the header declares %d" % (src_name, name, len(kinds), len(hkinds))
        assert kinds == hkinds, "%s: %s argument kinds %s, header %s" % (src_name, name, kinds, hkinds)
        assert ret == hret, "%s: %s returns %s, header %s" % (src_name, name, ret, hret)
    missing = HOST_PATH - set(decls)
    assert not missing, "%s does not bind the host-path functions %s" % (src_name, sorted(missing))


def test_header_parses_completely():
    from tfhe_aes_amd import _native

    h = header_decls()
    assert sorted(h) == _native.header_symbols()          # every declaration of the header was understood
    assert h["fheaes_create"] == ("i32", ["*params", "i32", "**ctx"])
    assert h["fheaes_clone_keys"] == ("i32", ["*ctx", "*ctx"])
    assert h["fheaes_last_error"] == ("*char", ["*ctx"])
    assert HOST_PATH <= set(h)


def test_rust_shim_extern_block_matches_the_header():
    text = (ROOT / "integration" / "rust_shim" / "src" / "lib.rs").read_text()
    assert "UNCOMPILED" in text.split("\n", 4)[1] + text.split("\n", 4)[2]      # the banner stays: these sources were never built
    _check("integration/rust_shim/src/lib.rs", rust_decls(text), header_decls())


def test_integration_md_bindings_match_the_header():
    text = (ROOT / "INTEGRATION.md").read_text()
    _check("INTEGRATION.md", rust_decls(text), header_decls())


def test_every_ffi_call_in_the_shim_is_declared_in_its_extern_block():
    text = (ROOT / "integration" / "rust_shim" / "src" / "lib.rs").read_text()
    declared = set(rust_decls(text))
    used = set(re.findall(r"\b(fheaes_[a-z0-9_]+)\s*\(", text)) - {"fheaes_params", "fheaes_ctx"}
    assert used <= declared | {"fheaes_params"}, "called but not declared: %s" % sorted(used - declared)
