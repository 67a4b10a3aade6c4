"""Bring the kernel sources to <= 160 columns without touching a token: comment lines are re-flowed, a trailing comment of an over-long code line moves onto its own line(s)
above the code, and what is still too long is cut at statement boundaries outside any bracket.  Preprocessor lines and continued lines are left alone.
The object code must come out identical (checked by the caller: md5 of the built library)."""
import re, sys, textwrap
W = 160


def split_comment(line):
    """index of a trailing // comment outside string / char literals, or -1"""
    i = 0; q = None
    while i < len(line) - 1:
        c = line[i]
        if q:
            if c == "\\": i += 2; continue
            if c == q: q = None
        elif c in "\"'": q = c
        elif c == "/" and line[i + 1] == "/": return i
        i += 1
    return -1


def wrap_comment(indent, text, lead=""):
    """`text` (the comment without its //) as comment lines of balanced length; `lead` = the blanks (and bullet) the text started with: continuation lines hang under it"""
    hang = " " * len(lead)
    if lead.strip():
        hang = " " * (len(lead) - len(lead.lstrip()) + 2)
    room = W - len(indent) - 3
    n = 1
    while True:      # the smallest number of lines that fits, then the narrowest width that still gives that number: no orphaned last words
        n_lines = textwrap.wrap(text, room - len(hang), break_long_words=False, break_on_hyphens=False)
        n = len(n_lines); break
    width = room - len(hang)
    for w in range(max(40, len(text) // max(1, n)), width + 1):
        t = textwrap.wrap(text, w, break_long_words=False, break_on_hyphens=False)
        if len(t) <= n:
            n_lines = t; break
    return [indent + "// " + (lead if k == 0 else hang) + t for k, t in enumerate(n_lines)] or [indent + "//"]


def cut_statements(indent, code):
    """cut `code` at '; ' where no bracket is open"""
    out = []; depth = 0; q = None; start = 0; i = 0
    while i < len(code):
        c = code[i]
        if q:
            if c == "\\": i += 2; continue
            if c == q: q = None
        elif c in "\"'": q = c
        elif c in "([{": depth += 1
        elif c in ")]}": depth -= 1
        elif c == ";" and depth == 0 and i + 1 < len(code) and code[i + 1] == " ":
            out.append(code[start:i + 1]); start = i + 2
        i += 1
    out.append(code[start:])
    out = [o for o in out if o.strip()]
    # greedily re-join pieces while they fit
    lines = []; cur = ""
    for o in out:
        if cur and len(indent) + len(cur) + 1 + len(o) <= W: cur += " " + o
        else:
            if cur: lines.append(indent + cur)
            cur = o
    if cur: lines.append(indent + cur)
    return lines


def process(path):
    src = open(path).read().split("\n"); out = []; changed = 0; left = 0
    for k, line in enumerate(src):
        prev_cont = k > 0 and src[k - 1].rstrip().endswith("\\")
        if len(line) <= W or line.lstrip().startswith("#") or line.rstrip().endswith("\\") or prev_cont:
            out.append(line); left += len(line) > W; continue
        indent = re.match(r"\s*", line).group(0)
        body = line[len(indent):]
        if body.startswith("//"):
            m = re.match(r"//\s?(\s*(?:[*-]\s+|\(\w\)\s+)?)", body)
            out += wrap_comment(indent, body[m.end():], m.group(1)); changed += 1; continue
        ci = split_comment(line)
        code = line if ci < 0 else line[:ci].rstrip()
        if ci >= 0:
            out += wrap_comment(indent, line[ci + 2:].strip())
        if len(code) > W:
            pieces = cut_statements(indent, code[len(indent):])
            out += pieces; left += sum(len(p) > W for p in pieces)
        else:
            out.append(code)
        changed += 1
    open(path, "w").write("\n".join(out))
    return changed, left


for p in sys.argv[1:]:
    c, l = process(p)
    print("%s: %d lines re-flowed, %d still longer than %d columns" % (p, c, l, W))
