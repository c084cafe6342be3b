#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2) rocpd SQLite database: per-kernel call count,
total/avg/min/max duration — the `--kernel-trace --stats` view, as text for profiles/."""
import sqlite3
import sys


def main(path, limit=40):
    db = sqlite3.connect(path)
    rows = db.execute("""select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start),
                                max(d.grid_size_x), max(d.grid_size_y), max(d.workgroup_size_x), max(s.arch_vgpr_count),
                                max(s.sgpr_count), max(d.group_segment_size), max(d.private_segment_size)
                         from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                         group by s.kernel_name order by 3 desc""").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("%-72s %6s %12s %10s %10s %10s %6s  %s" % ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "%", "grid x,y / wg / vgpr / sgpr / lds / scratch"))
    for r in rows[:limit]:
        name = r[0].replace(".kd", "")
        print("%-72s %6d %12.3f %10.3f %10.3f %10.3f %6.1f  %d,%d / %d / %d / %d / %d / %d" % (
            name[:72], r[1], r[2] / 1e6, r[2] / r[1] / 1e6, r[3] / 1e6, r[4] / 1e6, 100.0 * r[2] / total,
            r[5], r[6], r[7], r[8] or 0, r[9] or 0, r[10] or 0, r[11] or 0))
    if "--dispatches" in sys.argv:
        print("\n# every dispatch, in order")
        for r in db.execute("""select s.kernel_name, d.end-d.start, d.grid_size_x, d.grid_size_y from rocpd_kernel_dispatch d
                               join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"""):
            print("%-72s %10.3f ms  grid %d,%d" % (r[0].replace(".kd", "")[:72], r[1] / 1e6, r[2], r[3]))


if __name__ == "__main__":
    main(sys.argv[1])
