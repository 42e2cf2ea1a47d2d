"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace: per-kernel calls / total / average / min / max.
usage: python scripts/rocpd_stats.py <results.db> [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                      "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | sgpr | lds | grid_x | wg_x |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        name = r[0]
        if len(name) > 110:
            name = name[:107] + "..."
        lines.append("| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.1f | %d | %d | %d | %d | %d | %d |" % (
            name, r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total, r[6], r[7], r[8], r[9], r[10], r[11]))
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
