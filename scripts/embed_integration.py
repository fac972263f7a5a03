"""Refresh the verbatim copies of integration/*.h inside INTEGRATION.md (tests/test_integration_glue.py compares them byte for byte):
every ```cpp block whose first line is `// integration/<name> ...` is replaced by the file's current text.

  python scripts/embed_integration.py"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    path = os.path.join(ROOT, "INTEGRATION.md")
    md = open(path).read()
    done = []

    def swap(match):
        first = match.group(1).split("\n", 1)[0]
        m = re.match(r"// integration/([a-z_]+\.h)\b", first)
        if not m:
            return match.group(0)
        done.append(m.group(1))
        return "```cpp\n" + open(os.path.join(ROOT, "integration", m.group(1))).read().rstrip("\n") + "\n```"

    md = re.sub(r"```cpp\n(.*?)```", swap, md, flags=re.S)
    open(path, "w").write(md)
    print("refreshed:", ", ".join(done))


if __name__ == "__main__":
    main()
