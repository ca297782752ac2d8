"""rocprofv3 (ROCm 7.2) writes a rocpd sqlite database; this turns it into the small CSV summaries
committed under profiles/:   python tools/rocprof_summary.py <results.db> <out.csv> [--pmc]"""
import csv
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    if "--pmc" in sys.argv:
        # per kernel name: dispatches, mean counter value per dispatch
        # pmc_events already carries the kernel name of the dispatch (`name`) and the counter
        q = """select name as kernel, counter_name as counter, count(*) as dispatches, avg(counter_value) as mean_value,
                      sum(counter_value) as total_value
               from pmc_events group by name, counter_name order by total_value desc"""
        try:
            rows = c.execute(q).fetchall()
            cols = ["kernel", "counter", "dispatches", "mean_value_per_dispatch", "total_value"]
        except sqlite3.Error as e:  # schema differences between rocprofv3 builds: dump what is there
            cur = c.execute("select * from pmc_events limit 5")
            print("pmc_events columns:", [d[0] for d in cur.description], file=sys.stderr)
            cur = c.execute("select * from kernels limit 1")
            print("kernels columns:", [d[0] for d in cur.description], file=sys.stderr)
            raise SystemExit(str(e))
    else:
        cur = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
        rows = cur.fetchall()
        cols = ["kernel", "calls", "total_us", "avg_us", "percent"]
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(cols)
        for r in rows:
            w.writerow([round(x, 3) if isinstance(x, float) else x for x in r])
    print(f"{len(rows)} rows -> {out}")


if __name__ == "__main__":
    main()
