"""Development aid: print per-kernel averages of every counter in rocprofv3 rocpd databases."""
import sqlite3, sys
for db in sys.argv[1:]:
    cur = sqlite3.connect(db).cursor()
    for r in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                         "where kernel_name like '%points%' group by kernel_name, counter_name"):
        print("%-50s %-32s n=%d avg=%.6g" % (r[0][:50], r[1], r[2], r[3]))
