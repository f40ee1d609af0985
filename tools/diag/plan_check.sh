python tools/probe_gram.py 4213 65536 7 nocentre 2>&1 | tail -1
python tools/probe_gram.py 4213 65536 5 2>&1 | tail -1
python tools/probe_gram.py 4155 32768 7 nocentre 2>&1 | tail -1
python tools/probe_gram.py 16181 65536 3 nocentre 2>&1 | tail -1
python tools/probe_gram.py 411 10000 7 nocentre 2>&1 | tail -1
python tools/probe_gram.py 2000 65536 7 nocentre 2>&1 | tail -1
