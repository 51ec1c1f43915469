# closing fuzz campaign of a round on the current build: tools/gpu_fuzz_round4.sh [seed base] [scale]
S=${1:-4000}; K=${2:-1}
( timeout 2400 python tools/gpu_fuzz_orb.py $S $((700*K)) > gpurun_out/fuzz_orb_$S.log 2>&1; echo "orb rc=$?"; tail -2 gpurun_out/fuzz_orb_$S.log ) &
P1=$!
timeout 900 python tools/gpu_fuzz_misc.py $S $((300*K)) > gpurun_out/fuzz_misc_$S.log 2>&1; echo "misc rc=$?"; tail -1 gpurun_out/fuzz_misc_$S.log
timeout 900 python tools/gpu_fuzz_ba.py $S $((150*K)) > gpurun_out/fuzz_ba_$S.log 2>&1; echo "ba rc=$?"; tail -1 gpurun_out/fuzz_ba_$S.log
timeout 900 python tools/gpu_fuzz_loop.py $S $((200*K)) > gpurun_out/fuzz_loop_$S.log 2>&1; echo "loop rc=$?"; tail -1 gpurun_out/fuzz_loop_$S.log
timeout 1200 python tools/gpu_fuzz_lcd.py $S $((400*K)) > gpurun_out/fuzz_lcd_$S.log 2>&1; echo "lcd rc=$?"; tail -2 gpurun_out/fuzz_lcd_$S.log
wait $P1
