#!/usr/bin/env python3
"""TypeScript -> CommonJS type eraser for the reference's video valve graph (src/producer/mixer.ts, src/transitioner.ts,
src/combiner.ts, src/blackSilence.ts), so that the reference's OWN valves run under node 12 (there is no tsc in this image)
against mocks of `redioactive` and `beamcoder` and a recording mock of `nodencl`.

TEST INFRASTRUCTURE ONLY: output goes to oracle/_ref/work/js/ (git-ignored, never committed); only the golden *data* captured
from running it is committed (tests/golden/valve_trace.json).

A tokeniser plus a small type-expression parser - not a TypeScript compiler.  It erases what these files use: imports of
types, interfaces / type aliases, `implements`, member modifiers and `!`, parameter / variable / field / return annotations
(generic, tuple, union, function types, spanning lines), `as T`, `new X<T>()`, and rewrites `a?.b` to `__opt(a).b` (node 12 has
no optional chaining; __opt(null) is a proxy whose every property and call yields undefined)."""
import os
import re
import sys

REF = os.environ.get("PHANERON_REFERENCE", "/root/reference")

TOKEN = re.compile(r"""
    (?P<ws>[ \t\r\n]+)
  | (?P<line>//[^\n]*)
  | (?P<block>/\*.*?\*/)
  | (?P<str>'(?:\\.|[^'\\])*'|"(?:\\.|[^"\\])*")
  | (?P<num>0[xX][0-9a-fA-F]+|\d+\.?\d*(?:[eE][+-]?\d+)?|\.\d+)
  | (?P<id>[A-Za-z_$][\w$]*)
  | (?P<punct>=>|\.\.\.|\?\.|===|!==|==|!=|&&|\|\||\+\+|--|[{}()\[\];,<>+\-*/%&|^!~?:=.@])
""", re.S | re.X)


def tokenize(src):
    toks, i = [], 0
    while i < len(src):
        if src[i] == "`":  # template literal: one token, nested ${ } balanced
            j, depth = i + 1, 0
            while j < len(src):
                c = src[j]
                if c == "\\":
                    j += 2
                    continue
                if depth == 0 and c == "`":
                    break
                if c == "$" and src[j + 1] == "{":
                    depth += 1
                    j += 2
                    continue
                if depth and c == "{":
                    depth += 1
                if depth and c == "}":
                    depth -= 1
                j += 1
            toks.append(("str", src[i:j + 1]))
            i = j + 1
            continue
        m = TOKEN.match(src, i)
        if not m:
            raise SyntaxError("cannot tokenise at %r" % src[i:i + 40])
        kind = m.lastgroup
        toks.append((kind, m.group()))
        i = m.end()
    return toks


class Eraser:
    def __init__(self, toks):
        self.t = toks
        self.dead = [False] * len(toks)

    # -- navigation over significant tokens -------------------------------------------------------------------------
    def sig(self, i):
        return self.t[i][0] not in ("ws", "line", "block")

    def nxt(self, i):
        i += 1
        while i < len(self.t) and not self.sig(i):
            i += 1
        return i

    def prv(self, i):
        i -= 1
        while i >= 0 and not self.sig(i):
            i -= 1
        return i

    def val(self, i):
        return self.t[i][1] if 0 <= i < len(self.t) else None

    def kill(self, a, b):  # [a, b)
        for k in range(a, b):
            if self.t[k][0] not in ("line",):
                self.dead[k] = True

    def match(self, i):
        """index of the bracket closing the one at i"""
        pairs = {"(": ")", "[": "]", "{": "}", "<": ">"}
        o, c = self.val(i), pairs[self.val(i)]
        depth, j = 0, i
        while j < len(self.t):
            v = self.val(j)
            if self.sig(j):
                if v == o:
                    depth += 1
                elif v == c:
                    depth -= 1
                    if depth == 0:
                        return j
                elif o == "<" and v in (";", "{"):
                    raise SyntaxError("unbalanced <")
            j += 1
        raise SyntaxError("unbalanced %s" % o)

    # -- type expressions -------------------------------------------------------------------------------------------
    def parse_type(self, i):
        """i = first significant token of a type; returns the index of the first significant token AFTER it"""
        if self.val(i) in ("|", "&"):
            i = self.nxt(i)
        i = self.parse_primary(i)
        while self.val(i) in ("|", "&"):
            i = self.parse_primary(self.nxt(i))
        return i

    def parse_primary(self, i):
        v = self.val(i)
        if v in ("typeof", "keyof", "readonly"):
            return self.parse_primary(self.nxt(i))
        if v == "(":
            j = self.nxt(self.match(i))
            if self.val(j) == "=>":  # function type
                return self.parse_type(self.nxt(j))
            i = j
        elif v in ("{", "["):
            i = self.nxt(self.match(i))
        elif self.t[i][0] in ("str", "num"):
            i = self.nxt(i)
        elif self.t[i][0] == "id":
            i = self.nxt(i)
            while self.val(i) == "." and self.t[self.nxt(i)][0] == "id":
                i = self.nxt(self.nxt(i))
            if self.val(i) == "<":
                i = self.nxt(self.match(i))
        else:
            raise SyntaxError("not a type at %r" % "".join(x[1] for x in self.t[i:i + 8]))
        while self.val(i) == "[" and self.val(self.nxt(i)) == "]":  # T[]
            i = self.nxt(self.nxt(i))
        return i

    def erase_annotation(self, colon):
        """erase `: Type` starting at the colon; returns the index after"""
        end = self.parse_type(self.nxt(colon))
        # keep the whitespace that precedes the next token
        last = self.prv(end)
        self.kill(colon, last + 1)
        return end

    # -- passes -----------------------------------------------------------------------------------------------------
    def params(self, open_paren):
        """parameter list: drop `?` markers and annotations of top-level parameters"""
        close = self.match(open_paren)
        i = self.nxt(open_paren)
        depth = 0
        while i < close:
            v = self.val(i)
            if v in ("(", "[", "{"):
                j = self.match(i)
                if v == "(" and depth == 0 and self.is_function_parens(i):
                    self.function_at(i)
                i = self.nxt(j)
                continue
            if v == "?" and self.val(self.nxt(i)) == ":":
                self.dead[i] = True
            elif v == ":" and self.t[self.prv(i)][0] == "id" or (v == ":" and self.val(self.prv(i)) == "?"):
                i = self.erase_annotation(i)
                continue
            i = self.nxt(i)
        return close

    def is_function_parens(self, i):
        """does the ( at i open a parameter list?  What follows its ) decides: `=>`, `{` or a return annotation - unless a
        statement keyword stands before it"""
        before = self.val(self.prv(i))
        if before in ("if", "for", "while", "switch", "catch", "with", "return", "await", "typeof"):
            return False
        after = self.nxt(self.match(i))
        v = self.val(after)
        if v == "=>":
            return True
        if v == "{":  # method or function body: the token before ( is a name or `function`; a call before a block is `x(...) {`: not JS
            return self.t[self.prv(i)][0] == "id" and before not in ("if", "for", "while", "switch", "catch")
        if v == ":":
            try:
                end = self.parse_type(self.nxt(after))
            except SyntaxError:
                return False
            return self.val(end) in ("{", "=>")
        return False

    def function_at(self, open_paren):
        close = self.params(open_paren)
        after = self.nxt(close)
        if self.val(after) == ":":
            self.erase_annotation(after)

    def run(self):
        n = len(self.t)
        i = 0
        class_depth = []  # brace depths at which class bodies opened
        depth = 0
        while i < n:
            if not self.sig(i) or self.dead[i]:
                i += 1
                continue
            k, v = self.t[i]
            if v == "{":
                depth += 1
            elif v == "}":
                depth -= 1
                if class_depth and class_depth[-1] == depth + 1:
                    class_depth.pop()
            # interface / type alias (statement level)
            if v in ("interface", "type") and k == "id" and self.t[self.nxt(i)][0] == "id" and self.statement_start(i):
                start = i
                p = self.prv(i)
                if self.val(p) == "export":
                    start = p
                if v == "interface":
                    j = self.nxt(i)
                    while self.val(j) != "{":
                        j = self.nxt(j)
                    end = self.match(j) + 1
                else:
                    j = self.nxt(self.nxt(i))
                    if self.val(j) == "<":
                        j = self.nxt(self.match(j))
                    assert self.val(j) == "=", "type alias without ="
                    end = self.prv(self.parse_type(self.nxt(j))) + 1
                self.kill(start, end)
                i = end
                continue
            if v == "class" and k == "id":
                j = self.nxt(i)
                while self.val(j) != "{":
                    if self.val(j) == "implements":
                        e = j
                        while self.val(e) != "{":
                            e = self.nxt(e)
                        self.kill(j, self.prv(e) + 1)
                        j = e
                        break
                    j = self.nxt(j)
                class_depth.append(depth + 1)
                i += 1
                continue
            # class members
            if class_depth and class_depth[-1] == depth and k == "id" and self.member_start(i):
                j = i
                while self.val(j) in ("private", "protected", "public", "readonly", "abstract", "static", "async", "get", "set") and \
                        self.t[self.nxt(j)][0] in ("id",) :
                    if self.val(j) in ("private", "protected", "public", "readonly", "abstract"):
                        self.dead[j] = True
                        if j + 1 < n and self.t[j + 1][0] == "ws":
                            self.dead[j + 1] = True
                    j = self.nxt(j)
                name = j
                j = self.nxt(name)
                if self.val(j) in ("!", "?") and self.val(self.nxt(j)) == ":":
                    self.dead[j] = True
                    j = self.nxt(j)
                if self.val(j) == ":":
                    self.erase_annotation(j)
                elif self.val(j) == "(":
                    self.function_at(j)
                i = name + 1
                continue
            if v == "(" and self.is_function_parens(i):
                self.function_at(i)
                i += 1  # descend: nested arrows inside default values are rare; bodies are scanned as the loop goes on
                continue
            if v in ("const", "let", "var") and k == "id":
                j = self.nxt(i)
                if self.t[j][0] == "id" and self.val(self.nxt(j)) == ":":
                    self.erase_annotation(self.nxt(j))
            if v == "as" and k == "id" and self.t[self.prv(i)][0] in ("id", "str", "num") + () or \
                    (v == "as" and k == "id" and self.val(self.prv(i)) in (")", "]")):
                end = self.parse_type(self.nxt(i))
                start = i
                if self.t[i - 1][0] == "ws":
                    start = i - 1
                self.kill(start, self.prv(end) + 1)
                i = end
                continue
            if v == "new" and k == "id":
                j = self.nxt(i)
                while self.val(self.nxt(j)) == ".":
                    j = self.nxt(self.nxt(j))
                if self.val(self.nxt(j)) == "<":
                    a = self.nxt(j)
                    self.kill(a, self.match(a) + 1)
            if v == "!" and self.t[i - 1][0] != "ws" and (self.t[i - 1][0] == "id" or self.t[i - 1][1] in (")", "]")) and \
                    self.val(self.nxt(i)) in (".", ")", ",", ";", "[") and self.t[i - 1][1] not in ("return", "typeof", "await"):
                self.dead[i] = True  # non-null assertion
            i += 1
        return "".join(tok[1] for tok, d in zip(self.t, self.dead) if not d)

    def statement_start(self, i):
        p = self.prv(i)
        return p < 0 or self.val(p) in (";", "}", "{", "export") or "\n" in "".join(x[1] for x in self.t[p + 1:i])

    def member_start(self, i):
        p = self.prv(i)
        if self.val(p) not in (";", "}", "{") and "\n" not in "".join(x[1] for x in self.t[p + 1:i]):
            return False
        # a member starts a line inside the class body: modifiers, then name, then one of ! ? : ( =  or a line end
        j = i
        while self.val(j) in ("private", "protected", "public", "readonly", "abstract", "static", "async", "get", "set") and self.t[self.nxt(j)][0] == "id":
            j = self.nxt(j)
        return self.val(self.nxt(j)) in ("!", "?", ":", "(", "=") or j != i


def erase(src, modules):
    """modules: import source -> require() path (None: types only, the import disappears)"""
    def imp(m):
        what, mod = m.group(1).strip(), m.group(2)
        target = modules.get(mod, modules.get(mod.split("/")[-1]))
        if target is None:
            return ""
        mm = re.match(r"(\w+)?\s*,?\s*(\{[^}]*\})?", what, re.S)
        out = []
        if mm.group(1):
            out.append("const %s = require('%s').default" % (mm.group(1), target))
        if mm.group(2):
            names = re.sub(r"(\w+)\s+as\s+(\w+)", r"\1: \2", mm.group(2))
            out.append("const %s = require('%s')" % (" ".join(names.split()), target))
        return "\n".join(out)
    src = re.sub(r"^import\s+(.*?)\s+from\s+'([^']+)'\s*$", imp, src, flags=re.M | re.S)
    # optional chaining on a plain member chain (the only form these files use)
    src = re.sub(r"((?:this\.)?[A-Za-z_]\w*)\?\.", r"__opt(\1).", src)
    assert "?." not in src, "optional chaining in an unsupported position"
    names = []

    def exp_default(m):
        names.append("default:" + m.group(2))
        return "%s %s" % (m.group(1), m.group(2))

    def exp_named(m):
        names.append(m.group(2))
        return "%s %s" % (m.group(1), m.group(2))
    src = re.sub(r"^export\s+default\s+(class|function)\s+(\w+)", exp_default, src, flags=re.M)
    src = re.sub(r"^export\s+(class|function|const|async function)\s+(\w+)", exp_named, src, flags=re.M)
    out = Eraser(tokenize(src)).run()
    out = re.sub(r"^export\s*$", "", out, flags=re.M)  # `export` left behind by an erased interface / type
    tail = ["exports.default = %s" % n[8:] if n.startswith("default:") else "exports.%s = %s" % (n, n) for n in names]
    head = ("'use strict'\nconst __nothing = new Proxy(function () {}, { get: () => __nothing, apply: () => undefined })\n"
            "const __opt = (x) => (x === null || x === undefined ? __nothing : x)\n")
    return head + out + "\n" + "\n".join(tail) + "\n"


# file -> (output, import map additions); sources relative to src/
FILES = ["producer/mixer.ts", "transitioner.ts", "combiner.ts", "blackSilence.ts"]


def main(out_dir, mocks_dir):
    rel_mock = lambda frm, name: os.path.relpath(os.path.join(mocks_dir, name), os.path.dirname(os.path.join(out_dir, frm)))
    for rel in FILES:
        modules = {"nodencl": None, "events": "events", "redioactive": rel_mock(rel, "redioactive_mock.js"),
                   "beamcoder": rel_mock(rel, "beamcoder_mock.js"), "layer": None, "routeSource": None, "config": None}
        with open(os.path.join(REF, "src", rel)) as f:
            src = f.read()
        for m in re.finditer(r"from '(\.[^']+)'", src):  # local modules: already stripped by ts_strip.py / this script
            base = m.group(1).split("/")[-1]
            if base not in modules:
                modules[m.group(1)] = m.group(1)
        js = erase(src, modules)
        dst = os.path.join(out_dir, rel[:-3] + ".js")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        with open(dst, "w") as f:
            f.write(js)
    print("erased the types of %d files into %s" % (len(FILES), out_dir))


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "_ref", "work", "js"),
         sys.argv[2] if len(sys.argv) > 2 else os.path.join(here, "..", "..", "node", "test"))
