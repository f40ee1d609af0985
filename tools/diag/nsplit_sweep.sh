# Gram time against the sample-split factor and the workgroup order (steady-state DMA kernel, C2 shape)
for ns in 8 9 10 11 12 18 27; do VMCPDE_GRAM_NSPLIT=$ns python tools/probe_gram.py 4213 65536 7 nocentre 2>&1 | tail -1; done
for ns in 9 10 11; do VMCPDE_GRAM_ORDER=0 VMCPDE_GRAM_NSPLIT=$ns python tools/probe_gram.py 4213 65536 7 nocentre 2>&1 | tail -1; done
