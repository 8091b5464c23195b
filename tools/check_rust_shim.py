#!/usr/bin/env python
"""A no-rustc drift check of integration/rust/gpu.rs (the build image has no Rust toolchain).

What a compiler would reject first, checked here:
  1. brackets balance (comments, strings and character literals skipped);
  2. every `fn nqe_*` in the file's `extern "C"` blocks is declared in include/nqe.h with the same number of arguments, the same
     argument classes (i32 / i64 / u32 / u64 / usize / pointer-to-const / pointer-to-mut) and the same class of return value;
  3. every `Type::function(` on a type this file defines names a function some `impl … Type` block of this file defines;
  4. every `self.method(` inside an `impl … Type` block names a method of Type (any of its impl blocks, or a method of a trait
     the file implements for it whose signature the file itself declares), every `self.field` a field of Type;
  5. every method called on a GpuCtx (`ctx.x(`, `self.ctx.x(`) is a method of GpuCtx.

Usage: python tools/check_rust_shim.py [path]   → exit code 0 and "ok", or the list of problems.
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ lexing helpers
def strip_rust(src: str) -> str:
    """comments, string and char literals → spaces (newlines kept, so offsets and line numbers survive)"""
    out, i, n = [], 0, len(src)
    blank = lambda s: "".join(c if c == "\n" else " " for c in s)
    while i < n:
        two = src[i:i + 2]
        if two == "//":
            j = src.find("\n", i)
            j = n if j < 0 else j
            out.append(blank(src[i:j]))
            i = j
        elif two == "/*":
            depth, j = 1, i + 2
            while j < n and depth:
                if src[j:j + 2] == "/*":
                    depth, j = depth + 1, j + 2
                elif src[j:j + 2] == "*/":
                    depth, j = depth - 1, j + 2
                else:
                    j += 1
            out.append(blank(src[i:j]))
            i = j
        elif src[i] == '"':
            j = i + 1
            while j < n and src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            # the ABI string of an extern block is kept: the block parser looks for it
            lit = src[i:j + 1]
            out.append(lit if lit == '"C"' else '"' + blank(src[i + 1:j]) + '"')
            i = j + 1
        elif src[i] == "'":
            m = re.match(r"'(\\.|[^\\'])'", src[i:])
            if m:  # a character literal (a lifetime has no closing quote)
                out.append(" " * len(m.group(0)))
                i += len(m.group(0))
            else:
                out.append(src[i])
                i += 1
        else:
            out.append(src[i])
            i += 1
    return "".join(out)


def check_brackets(code: str, problems: list) -> None:
    pairs, stack = {")": "(", "]": "[", "}": "{"}, []
    line = 1
    for ch in code:
        if ch == "\n":
            line += 1
        elif ch in "([{":
            stack.append((ch, line))
        elif ch in ")]}":
            if not stack or stack[-1][0] != pairs[ch]:
                problems.append(f"line {line}: unbalanced '{ch}'")
                return
            stack.pop()
    if stack:
        problems.append(f"line {stack[-1][1]}: '{stack[-1][0]}' is never closed")


def matching(code: str, open_at: int) -> int:
    """index of the bracket that closes the one at `open_at`"""
    o = code[open_at]
    c = {"(": ")", "[": "]", "{": "}", "<": ">"}[o]
    depth = 0
    for k in range(open_at, len(code)):
        if code[k] == o:
            depth += 1
        elif code[k] == c:
            depth -= 1
            if depth == 0:
                return k
    return -1


def split_top(s: str) -> list:
    """split at commas outside any bracket"""
    parts, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
    tail = "".join(cur).strip()
    if tail:
        parts.append(tail)
    return parts


def line_of(code: str, pos: int) -> int:
    return code.count("\n", 0, pos) + 1


# ------------------------------------------------------------------ 2. the FFI declarations against the header
def c_class(t: str) -> str:
    t = " ".join(t.replace("*", " * ").split())
    if "*" in t:
        before_last = t[:t.rfind("*")].strip()
        stars = t.count("*")
        const_pointee = before_last.endswith("const") or (stars == 1 and before_last.startswith("const"))
        return "ptr_const" if const_pointee else "ptr_mut"
    base = t.replace("const", "").strip()
    return {"int32_t": "i32", "int64_t": "i64", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "double": "f64", "nqe_status": "i32",
            "void": "void"}.get(base, base)


def header_functions() -> dict:
    src = open(os.path.join(ROOT, "include", "nqe.h")).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    fns = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ ]*?[ \*]+)\b(nqe_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        if "typedef" in ret or "(*" in m.group(0)[:m.group(0).find(name)]:
            continue
        arglist = [] if args.strip() in ("", "void") else split_top(args)
        classes = []
        for a in arglist:
            a = a.strip()
            # drop the parameter name (the last identifier), keep the type
            mm = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)\s*$", a, flags=re.S)
            classes.append(c_class(mm.group(1) if mm and ("*" in a or " " in a.strip()) else a))
        fns[name] = (c_class(ret), classes)
    return fns


def rust_class(t: str) -> str:
    t = t.strip()
    if t.startswith("*const"):
        return "ptr_const"
    if t.startswith("*mut"):
        return "ptr_mut"
    return t


def extern_functions(code: str) -> dict:
    fns = {}
    for m in re.finditer(r'extern\s+"C"\s*\{', code):
        end = matching(code, m.end() - 1)
        body = code[m.end():end]
        for f in re.finditer(r"\bfn\s+([a-z0-9_]+)\s*\(", body):
            close = matching(body, f.end() - 1)
            args = split_top(body[f.end():close])
            rest = body[close + 1:body.find(";", close)]
            ret = rust_class(rest.split("->", 1)[1]) if "->" in rest else "void"
            fns[f.group(1)] = (ret, [rust_class(a.split(":", 1)[1]) for a in args], line_of(code, m.end() + f.start()))
    return fns


def check_ffi(code: str, problems: list) -> int:
    header, rust = header_functions(), extern_functions(code)
    for name, (ret, args, line) in sorted(rust.items()):
        if name not in header:
            problems.append(f"line {line}: extern fn {name} is not declared in include/nqe.h")
            continue
        hret, hargs = header[name]
        if len(args) != len(hargs):
            problems.append(f"line {line}: {name} takes {len(hargs)} arguments in nqe.h, {len(args)} here")
            continue
        for k, (a, h) in enumerate(zip(args, hargs)):
            if a != h:
                problems.append(f"line {line}: {name} argument {k + 1} is {h} in nqe.h, {a} here")
        if ret != hret:
            problems.append(f"line {line}: {name} returns {hret} in nqe.h, {ret} here")
    # every nqe_* symbol the code CALLS must be declared in an extern block
    for m in re.finditer(r"\b(nqe_[a-z0-9_]+)\s*\(", code):
        if m.group(1) not in rust:
            problems.append(f"line {line_of(code, m.start())}: {m.group(1)} is called but not declared in an extern block")
    return len(rust)


# ------------------------------------------------------------------ 3-5. the file's own types
def own_items(code: str):
    """→ (types: {name: set(fields)}, methods: {type: set(fn names)}, impl spans [(type, start, end)], trait methods {trait: set})"""
    types, methods, spans, traits = {}, {}, [], {}
    for m in re.finditer(r"\b(?:pub\s+)?(struct|enum|union)\s+([A-Z][A-Za-z0-9]*)\s*([\({;])", code):
        name, opener = m.group(2), m.group(3)
        fields = set()
        if opener == "{":
            end = matching(code, m.end() - 1)
            for part in split_top(code[m.end():end]):
                fm = re.match(r"(?:pub(?:\([a-z]+\))?\s+)?([a-z_][a-z0-9_]*)\s*:", part.strip())
                if fm:
                    fields.add(fm.group(1))
        elif opener == "(":
            end = matching(code, m.end() - 1)
            fields = {str(k) for k in range(len(split_top(code[m.end():end])))}
        types[name] = fields
    for m in re.finditer(r"\b(?:pub\s+)?trait\s+([A-Z][A-Za-z0-9]*)[^{]*\{", code):
        end = matching(code, m.end() - 1)
        traits[m.group(1)] = set(re.findall(r"\bfn\s+([a-z_][a-z0-9_]*)", code[m.end():end]))
    for m in re.finditer(r"\bimpl\s+(?:([A-Z][A-Za-z0-9]*)\s+for\s+)?([A-Z][A-Za-z0-9]*)\s*\{", code):
        trait, ty = m.group(1), m.group(2)
        end = matching(code, m.end() - 1)
        spans.append((ty, m.end(), end))
        fns = set(re.findall(r"\bfn\s+([a-z_][a-z0-9_]*)", code[m.end():end]))
        methods.setdefault(ty, set()).update(fns)
        if trait in traits:  # default methods of the file's own traits
            methods[ty].update(traits[trait])
    return types, methods, spans, traits


# methods of the reference's traits that the Gpu* operators inherit or that std provides on every value
ALWAYS = {"clone", "as_ref", "as_ptr", "as_mut_ptr", "len", "is_empty", "iter", "to_string", "into", "unwrap", "is_some", "is_none", "map", "lock", "first",
          "get", "cloned"}


def check_own_calls(code: str, problems: list) -> int:
    types, methods, spans, traits = own_items(code)
    checked = 0
    for m in re.finditer(r"\b([A-Z][A-Za-z0-9]*)::([a-z_][a-z0-9_]*)\s*\(", code):
        ty, fn = m.group(1), m.group(2)
        if ty in types and ty in methods or ty in types:
            checked += 1
            if fn not in methods.get(ty, set()):
                problems.append(f"line {line_of(code, m.start())}: {ty}::{fn} is called but no impl of {ty} in this file defines it")
    for ty, start, end in spans:
        body = code[start:end]
        for m in re.finditer(r"\bself\.([a-z_0-9][a-z0-9_]*)\s*(\()?", body):
            name, call = m.group(1), m.group(2)
            checked += 1
            where = f"line {line_of(code, start + m.start())}"
            if call:
                if name not in methods.get(ty, set()) and name not in ALWAYS:
                    problems.append(f"{where}: self.{name}() inside impl {ty}, which defines no such method")
            elif ty in types and name not in types[ty]:
                problems.append(f"{where}: self.{name} inside impl {ty}, which has no such field")
    ctx_methods = methods.get("GpuCtx", set())
    for m in re.finditer(r"\b(?:self\.)?ctx\.([a-z_][a-z0-9_]*)\s*\(", code):
        checked += 1
        if m.group(1) not in ctx_methods and m.group(1) not in ALWAYS:
            problems.append(f"line {line_of(code, m.start())}: ctx.{m.group(1)}() is not a method of GpuCtx")
    return checked


def check(path: str):
    src = open(path).read()
    code = strip_rust(src)
    problems: list = []
    check_brackets(code, problems)
    n_ffi = check_ffi(code, problems) if not problems else 0
    n_calls = check_own_calls(code, problems) if not problems else 0
    return problems, n_ffi, n_calls


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "integration", "rust", "gpu.rs")
    problems, n_ffi, n_calls = check(path)
    if problems:
        print("\n".join(problems))
        sys.exit(1)
    print(f"ok: {n_ffi} extern declarations match include/nqe.h, {n_calls} calls / field accesses on the file's own types resolve")


if __name__ == "__main__":
    main()
