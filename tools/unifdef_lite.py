#!/usr/bin/env python3
"""Removes the preprocessor branches of macros that are known to be UNDEFINED from source files (a minimal unifdef):
    tools/unifdef_lite.py -U MACRO [-U MACRO ...] file...
Handles #ifdef / #ifndef / #if with defined() || && ! / #elif / #else / #endif; a directive that mentions any macro not listed is kept."""
import re, sys

def evaluate(expr, undef):
    """True / False when the expression only mentions known-undefined macros, else None"""
    names = set(re.findall(r"defined\s*\(\s*(\w+)\s*\)", expr))
    if not names or not names <= undef:
        return None
    py = re.sub(r"defined\s*\(\s*\w+\s*\)", "False", expr)
    py = py.replace("||", " or ").replace("&&", " and ")
    py = re.sub(r"!(?!=)", " not ", py)
    if re.search(r"[^\sA-Za-z()]", py):
        return None
    try:
        return bool(eval(py))
    except Exception:
        return None

def process(text, undef):
    out = []
    stack = []  # entries: dict(known, taken, active, emitted_any)
    def active():
        return all(s["active"] for s in stack)
    for line in text.split("\n"):
        m = re.match(r"^\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)$", line)
        if not m:
            if active():
                out.append(line)
            continue
        kind, rest = m.group(1), m.group(2)
        rest_nc = re.sub(r"//.*$", "", rest).strip()
        if kind in ("ifdef", "ifndef", "if"):
            if kind == "ifdef":
                name = rest_nc.split()[0] if rest_nc else ""
                val = False if name in undef else None
            elif kind == "ifndef":
                name = rest_nc.split()[0] if rest_nc else ""
                val = True if name in undef else None
            else:
                val = evaluate(rest_nc, undef)
            parent = active()
            if val is None:
                stack.append({"known": False, "active": True, "parent": parent})
                if parent:
                    out.append(line)
            else:
                stack.append({"known": True, "active": val, "taken": val, "parent": parent})
        elif kind == "elif":
            s = stack[-1]
            if not s["known"]:
                if all(t["active"] for t in stack[:-1]):
                    out.append(line)
            else:
                val = evaluate(rest_nc, undef)
                if s["taken"]:
                    s["active"] = False
                elif val is None:
                    raise SystemExit("unifdef_lite: #elif with unknown macros after a known #if: " + line)
                else:
                    s["active"] = val
                    s["taken"] = val
        elif kind == "else":
            s = stack[-1]
            if not s["known"]:
                if all(t["active"] for t in stack[:-1]):
                    out.append(line)
            else:
                s["active"] = not s["taken"]
                s["taken"] = True
        else:  # endif
            s = stack.pop()
            if not s["known"] and active():
                out.append(line)
    if stack:
        raise SystemExit("unifdef_lite: unbalanced directives")
    return "\n".join(out)

def main():
    undef, files = set(), []
    a = sys.argv[1:]
    while a:
        if a[0] == "-U":
            undef.add(a[1]); a = a[2:]
        else:
            files.append(a[0]); a = a[1:]
    for f in files:
        t = open(f).read()
        n = process(t, undef)
        if n != t:
            open(f, "w").write(n)
            print(f"{f}: {t.count(chr(10))} -> {n.count(chr(10))} lines")

if __name__ == "__main__":
    main()
