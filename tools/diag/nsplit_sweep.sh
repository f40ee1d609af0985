# Gram time against the sample-split factor and the workgroup order (steady-state DMA kernel)
for o in 1 2; do for ns in 8 9 10 11; do VMCPDE_GRAM_ORDER=$o VMCPDE_GRAM_NSPLIT=$ns python tools/probe_gram.py 4213 65536 7 nocentre 2>&1 | tail -1; done; done
for o in 1 2; do VMCPDE_GRAM_ORDER=$o python tools/probe_gram.py 4213 65536 7 nocentre 2>&1 | tail -1; done
for o in 1 2; do VMCPDE_GRAM_ORDER=$o python tools/probe_gram.py 4155 32768 7 nocentre 2>&1 | tail -1; done
for o in 1 2; do VMCPDE_GRAM_ORDER=$o python tools/probe_gram.py 16181 65536 3 nocentre 2>&1 | tail -1; done
for o in 1 2; do VMCPDE_GRAM_ORDER=$o python tools/probe_gram.py 411 10000 7 nocentre 2>&1 | tail -1; done
