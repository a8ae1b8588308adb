"""Summarise a rocprofv3 (ROCm 7.x) rocpd sqlite database into the per-kernel table `--stats` prints:
name, calls, total/avg/min/max duration.   python tools/rocpd_stats.py <results.db> [out.md]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by %s order by sum(end-start) desc" % (name_col, name_col)).fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, c, s, a, mn, mx in rows:
        short = n if len(n) < 110 else n[:107] + "..."
        lines.append("| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.1f |" % (short, c, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3,
                                                                        100.0 * s / tot))
    out = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")
    print(out)


if __name__ == "__main__":
    main()
