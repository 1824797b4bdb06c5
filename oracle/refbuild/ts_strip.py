#!/usr/bin/env python3
"""Minimal TypeScript -> CommonJS type stripper for the reference's src/process/*.ts and
src/clJobQueue.ts, so the reference's OWN host code can be executed under node 12 (there
is no tsc in this image) against a recording mock of `nodencl`.

TEST INFRASTRUCTURE ONLY: output goes to oracle/_ref/work/js/ (git-ignored, never committed);
only golden *data* captured from running it is committed (tests/golden/).

It is not a general TS compiler - it handles exactly the constructs those files use:
type-only imports, interfaces / type aliases, member modifiers, parameter / return /
field annotations, `as T` casts, `abstract` members, enums, optional chaining.
"""
import os
import re
import sys

REF = os.environ.get("PHANERON_REFERENCE", "/root/reference")

TYPE_ATOM = (
    r"(?:number|string|boolean|void|undefined|null|any|Buffer|Float32Array|Uint32Array|"
    r"nodenCLContext|OpenCLBuffer|OpenCLProgram|KernelParams|RunTimings|ImageDims|"
    r"ClJobs|ClProcessJobs|ClJob|JobID|JobCB|JobsRequest|PackImpl|ProcessImpl|ImageProcess|"
    r"Interlace|ColParams|ColParam|YadifConfig|YadifMode|Loader|Saver|EventEmitter|"
    r"\[number, number\]|number\[\]|"
    r"\(\) => void)"
)
TYPE_ONE = r"(?:(?:Array|Promise|Map)<[^<>]*(?:<[^<>]*>)?[^<>]*>|%s(?:\[\])*)" % TYPE_ATOM
TYPE = r"%s(?:\s*\|\s*%s)*" % (TYPE_ONE, TYPE_ONE)


def strip(src, local_modules):
    # comments first (block comments contain ':' patterns in prose)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    # interfaces and type aliases
    src = re.sub(r"^(?:export\s+)?interface\s+\w+\s*\{.*?^\}", "", src, flags=re.S | re.M)
    src = re.sub(r"^(?:export\s+)?type\s+\w+\s*=\s*\{[^}]*\}\s*$", "", src, flags=re.M)
    src = re.sub(r"^(?://[^\n]*\n)?(?:export\s+)?type\s+\w+\s*=[^\n{]*$", "", src, flags=re.M)
    # enums -> frozen objects
    def enum(m):
        body = ", ".join(
            "%s: %s" % tuple(p.strip() for p in item.split("="))
            for item in m.group(2).split(",")
            if item.strip()
        )
        return "const %s = Object.freeze({ %s })\nexports.%s = %s" % (m.group(1), body, m.group(1), m.group(1))
    src = re.sub(r"export\s+enum\s+(\w+)\s*\{(.*?)\}", enum, src, flags=re.S)

    # imports
    def imp(m):
        what, mod = m.group(1).strip(), m.group(2)
        base = mod.split("/")[-1]
        if mod == "events":
            return "const %s = require('events')" % what
        if base not in local_modules:
            return ""  # nodencl etc: types only
        default, named = None, None
        mm = re.match(r"(\w+)?\s*,?\s*(\{[^}]*\})?", what)
        default, named = mm.group(1), mm.group(2)
        out = []
        if default:
            out.append("const %s = require('%s').default" % (default, mod))
        if named:
            out.append("const %s = require('%s')" % (named, mod))
        return "\n".join(out)
    src = re.sub(r"^import\s+(.*?)\s+from\s+'([^']+)'\s*$", imp, src, flags=re.M | re.S)

    # abstract members disappear, modifiers are dropped
    src = re.sub(r"^\s*(?:protected\s+|public\s+)?abstract\s+(?!class)[^\n]*$", "", src, flags=re.M)
    src = re.sub(r"\b(?:private|protected|public|readonly|abstract)\s+", "", src)
    # casts
    src = re.sub(r"\s+as\s+%s" % TYPE, "", src)
    # generic call / constructor type arguments
    src = re.sub(r"new (Map|Array|Promise)<[^<>()]*(?:<[^<>]*>)?[^<>()]*>\(", r"new \1(", src)
    # return-type annotations:  ): T {   ): T =>
    src = re.sub(r"\)\s*:\s*%s\s*(\{|=>)" % TYPE, r") \1", src)
    # parameter / variable / field annotations:  name?: T
    src = re.sub(r"(\b\w+)\??\s*:\s*%s(?=\s*[,)=;\n])" % TYPE, r"\1", src)
    # optional chaining (node 12 has none): the path's files only use the statement form
    # `this.field?.method(...)`, which is rewritten to a guarded call with the same semantics
    src = re.sub(r"^(\s*)((?:this\.)?\w+)\?\.(\w+\([^\n]*\))\s*$", r"\1if (\2 != null) \2.\3", src, flags=re.M)
    assert "?." not in src, "optional chaining in an unsupported position"
    # V8 7.8 rejects a class field literally named `in`
    src = re.sub(r"^(\s*)in = ", r"\1;['in'] = ", src, flags=re.M)

    # exports
    names = []
    def exp_named(m):
        names.append(m.group(2))
        return "%s %s" % (m.group(1), m.group(2))
    src = re.sub(r"^export\s+default\s+(class|function)\s+(\w+)",
                 lambda m: (names.append("default:" + m.group(2)) or "%s %s" % (m.group(1), m.group(2))),
                 src, flags=re.M)
    src = re.sub(r"^export\s+(class|function|const)\s+(\w+)", exp_named, src, flags=re.M)
    tail = []
    for n in names:
        if n.startswith("default:"):
            tail.append("exports.default = %s" % n[8:])
        else:
            tail.append("exports.%s = %s" % (n, n))
    return "'use strict'\n" + src + "\n" + "\n".join(tail) + "\n"


FILES = [
    "clJobQueue.ts",
    "process/colourMaths.ts",
    "process/packer.ts",
    "process/imageProcess.ts",
    "process/loadSave.ts",
    "process/io.ts",
    "process/v210.ts",
    "process/yuv422p10.ts",
    "process/yuv422p8.ts",
    "process/yuv420p.ts",
    "process/nv12.ts",
    "process/rgba8.ts",
    "process/bgra8.ts",
    "process/yadifCl.ts",
    "process/yadif.ts",
    "process/transform.ts",
    "process/resize.ts",
    "process/combine.ts",
    "process/transition.ts",
    "process/mix.ts",
    "process/wipe.ts",
]


def main(out_dir):
    local = {os.path.splitext(os.path.basename(f))[0] for f in FILES}
    for rel in FILES:
        with open(os.path.join(REF, "src", rel)) as f:
            js = strip(f.read(), local)
        dst = os.path.join(out_dir, rel[:-3] + ".js")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        with open(dst, "w") as f:
            f.write(js)
    print("stripped %d files into %s" % (len(FILES), out_dir))


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "_ref", "work", "js"))
