# closing fuzz campaign of a round on the current build: tools/gpu_fuzz_round4.sh [seed base]
S=${1:-4000}
( timeout 1500 python tools/gpu_fuzz_orb.py $S 700 > gpurun_out/fuzz_orb_$S.log 2>&1; echo "orb rc=$?"; tail -2 gpurun_out/fuzz_orb_$S.log ) &
P1=$!
timeout 600 python tools/gpu_fuzz_misc.py $S 300 > gpurun_out/fuzz_misc_$S.log 2>&1; echo "misc rc=$?"; tail -1 gpurun_out/fuzz_misc_$S.log
timeout 600 python tools/gpu_fuzz_ba.py $S 150 > gpurun_out/fuzz_ba_$S.log 2>&1; echo "ba rc=$?"; tail -1 gpurun_out/fuzz_ba_$S.log
timeout 600 python tools/gpu_fuzz_loop.py $S 200 > gpurun_out/fuzz_loop_$S.log 2>&1; echo "loop rc=$?"; tail -1 gpurun_out/fuzz_loop_$S.log
timeout 900 python tools/gpu_fuzz_lcd.py $S 400 > gpurun_out/fuzz_lcd_$S.log 2>&1; echo "lcd rc=$?"; tail -2 gpurun_out/fuzz_lcd_$S.log
wait $P1
