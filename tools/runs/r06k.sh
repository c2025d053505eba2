# soaks on round 6's final build: the overlapped native loop (chain-first order) against the oracle's replay; the arch5
# engine (owed tails, publisher flush, one-launch trainer) on tiny rings with several samplers and trainers
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06k
( timeout -k 10 400 python3 -u tools/soak_overlapped.py --kind khop2 --fanout 15,10,5 --rounds 20 --per-round 48 > gpurun_out/r06k/soak_a.log 2>&1; echo "a rc=$?"; tail -2 gpurun_out/r06k/soak_a.log ) 
( timeout -k 10 300 python3 -u tools/soak_overlapped.py --kind khop2 --rounds 12 --per-round 48 --help-after 0 > gpurun_out/r06k/soak_b.log 2>&1; echo "b rc=$?"; tail -1 gpurun_out/r06k/soak_b.log )
( timeout -k 10 300 python3 -u tools/soak_overlapped.py --kind khop2 --streams 4 --per-round 50 --batch 3000 --rounds 12 > gpurun_out/r06k/soak_c.log 2>&1; echo "c rc=$?"; tail -1 gpurun_out/r06k/soak_c.log )
( timeout -k 10 300 python3 -u tools/soak_overlapped.py --kind weighted_khop_prefix --fanout 5,10,15 --rounds 6 --per-round 48 > gpurun_out/r06k/soak_d.log 2>&1; echo "d rc=$?"; tail -1 gpurun_out/r06k/soak_d.log )
bash tools/soak_queue.sh > gpurun_out/r06k/soak_queue.txt 2>&1; echo "queue rc=$?"; cat gpurun_out/r06k/soak_queue.txt
