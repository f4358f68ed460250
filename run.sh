make
./simulation.out
# the reference then starts its viewer (run.sh:3-4 there); its GUI reads ./data/*.bin unchanged:
# python /path/to/reference/GUI/main.py
