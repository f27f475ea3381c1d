import sqlite3,re,sys
db=sqlite3.connect(sys.argv[1])
rows=db.execute("select name,start,end,grid_x,workgroup_x from kernels order by start").fetchall()
starts=[s for n,s,e,g,w in rows if 'stem_stats_kernel' in n]
k=len(starts)-3
a,b=starts[k],starts[k+1]
print("step %.3f ms"%((b-a)/1e6))
for n,s,e,g,w in rows:
    if a<=s<b and ('nt_kernel' in n):
        nm=re.sub(r"\(.*","",n).replace("_ZN5frhip","")[:40]
        print("%-42s blocks=%6d  %7.1f us  t=%.2f ms"%(nm,g//w,(e-s)/1e3,(s-a)/1e6))
