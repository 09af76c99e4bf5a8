"""Prints the layered engine's schedule for config C2 (SAMD_LY_DUMP=1: steps, items per wave, estimated cycles)."""
import os, sys
os.environ["SAMD_LY_DUMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sionna_amd.phy as phy
enc = phy.fec.ldpc.LDPC5GEncoder(2816, 8448, num_bits_per_symbol=6, bg="bg1")
dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", cn_schedule="layered", num_iter=1)
enc._handle(dec._nb_pruned_nodes)
