"""Per-kernel totals from a rocprofv3 rocpd database (the default output format): name, calls, total ms, mean us."""
import sqlite3
import sys


def stats(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    rows = db.execute(f"select s.kernel_name, count(*), sum(d.end - d.start) from {disp} d join {sym} s on d.kernel_id = s.id "
                      "group by s.kernel_name order by 3 desc").fetchall()
    return rows


if __name__ == "__main__":
    rows = stats(sys.argv[1])
    tot = sum(r[2] for r in rows)
    for name, n, ns in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
        print(f"{ns / 1e6:10.3f} ms {100 * ns / tot:5.1f}% {n:6d} x {ns / n / 1e3:9.1f} us  {name[:110]}")
