#!/usr/bin/env python3
"""Resolve the preprocessor conditionals of a source file that depend ONLY on the macros given on the command line
(NAME=VALUE: defined with that value, NAME=: undefined) and drop their `#ifndef NAME / #define NAME v / #endif` default blocks;
every other line, including conditionals on other macros, is left as it is. Used once per retired lab knob:
    tools/dev/unifdef.py file.hip ED2_SKIP=0 ED2_T1_LDS=0 ED2_WAVES_PER_EU= > file.new
"""
import re, sys

def main():
    path, known = sys.argv[1], {}
    for a in sys.argv[2:]:
        k, _, v = a.partition("=")
        known[k] = v if v != "" else None
    ident = re.compile(r"[A-Za-z_]\w*")

    def evaluate(expr):
        """value of a #if expression, or None when it mentions a macro we do not know"""
        e = re.sub(r"/\*.*?\*/", "", expr).strip()
        def sub_defined(m):
            n = m.group(1) or m.group(2)
            if n not in known: raise KeyError(n)
            return "1" if known[n] is not None else "0"
        try:
            e = re.sub(r"defined\s*\(\s*(\w+)\s*\)|defined\s+(\w+)", sub_defined, e)
            def sub_id(m):
                n = m.group(0)
                if n in ("and", "or", "not"): return n
                if n not in known: raise KeyError(n)
                return known[n] if known[n] is not None else "0"
            e = e.replace("&&", " and ").replace("||", " or ")
            e = re.sub(r"!(?!=)", " not ", e)
            e = ident.sub(sub_id, e)
            return bool(eval(e, {"__builtins__": {}}))
        except (KeyError, SyntaxError):
            return None

    lines = open(path).read().split("\n")
    out, stack = [], []   # stack entries: dict(kind="known"/"unknown", taken=bool, active=bool)
    def emitting():
        return all(f["active"] for f in stack)
    i = 0
    while i < len(lines):
        ln = lines[i]
        s = ln.strip()
        m = re.match(r"#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)", s)
        if not m:
            if emitting(): out.append(ln)
            i += 1
            continue
        d, rest = m.group(1), m.group(2).strip()
        if d in ("if", "ifdef", "ifndef"):
            if d == "if": v = evaluate(rest)
            else:
                n = re.sub(r"/\*.*", "", rest).strip()
                v = None if n not in known else ((known[n] is not None) == (d == "ifdef"))
            # `#ifndef KNOWN` / `#define KNOWN ...` / `#endif`: the default block of a retired knob -- drop it whole
            if d == "ifndef" and v is not None and i + 2 < len(lines) and re.match(r"#\s*define\s+" + re.escape(n) + r"\b", lines[i + 1].strip()):
                j = i + 2
                while j < len(lines) and not lines[j].strip().startswith("#endif"): j += 1   # comment continuation lines
                i = j + 1
                continue
            if v is None:
                if emitting(): out.append(ln)
                stack.append(dict(kind="unknown", active=True))
            else:
                stack.append(dict(kind="known", active=v, taken=v))
        elif d == "elif":
            f = stack[-1]
            if f["kind"] == "unknown":
                if all(g["active"] for g in stack[:-1]): out.append(ln)
            else:
                v = evaluate(rest)
                if v is None: raise SystemExit("%s:%d: #elif on unknown macros under a resolved #if: resolve by hand" % (path, i + 1))
                f["active"] = (not f["taken"]) and v
                f["taken"] = f["taken"] or v
        elif d == "else":
            f = stack[-1]
            if f["kind"] == "unknown":
                if all(g["active"] for g in stack[:-1]): out.append(ln)
            else:
                f["active"] = not f["taken"]
                f["taken"] = True
        else:
            f = stack.pop()
            if f["kind"] == "unknown" and emitting(): out.append(ln)
        i += 1
    sys.stdout.write("\n".join(out))

if __name__ == "__main__":
    main()
