"""Re-flow the prose of a Markdown file to a maximum line width (tables, code fences, headings and HTML are left alone;
list items keep their hanging indent).  Used once to make DESIGN.md and docs/*.md auditable: python scripts/wrap_md.py FILE [WIDTH]"""
import re
import sys
import textwrap


def wrap(text: str, width: int = 118) -> str:
    out, para, in_code = [], [], False

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*)([*\-+]|\d+\.)\s+", first)
        if m:
            ind, sub = m.group(1) + m.group(2) + " ", " " * (len(m.group(1)) + len(m.group(2)) + 1)
            body = " ".join([first[m.end():]] + [l.strip() for l in para[1:]])
        else:
            lead = re.match(r"^\s*", first).group(0)
            ind = sub = lead
            body = " ".join(l.strip() for l in para)
        out.extend(textwrap.wrap(body, width, initial_indent=ind, subsequent_indent=sub, break_long_words=False, break_on_hyphens=False))
        para.clear()

    for line in text.split("\n"):
        s = line.strip()
        if s.startswith("```"):
            flush(); in_code = not in_code; out.append(line); continue
        if in_code or s.startswith("|") or s.startswith("#") or s.startswith(">") or s == "" or s.startswith("<"):
            flush(); out.append(line); continue
        if re.match(r"^\s*([*\-+]|\d+\.)\s+", line) and para:       # a new list item ends the previous one
            flush()
        para.append(line)
    flush()
    return "\n".join(out)


if __name__ == "__main__":
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 118
    src = open(path).read()
    open(path, "w").write(wrap(src, width))
