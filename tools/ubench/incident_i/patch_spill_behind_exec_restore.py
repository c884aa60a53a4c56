"""python patch_spill_behind_exec_restore.py bis/lib_135352.so: the causal test of incident (i).  The 24 bytes

    v_accvgpr_write_b32 a20, v235 ; v_accvgpr_write_b32 a14, v234 ; s_mov_b64 s[78:79], s[60:61] ; s_or_b64 exec, exec, s[0:1]

at the head of the join block (one occurrence in the library) are reordered in place to

    s_mov_b64 s[78:79], s[60:61] ; s_or_b64 exec, exec, s[0:1] ; v_accvgpr_write_b32 a20, v235 ; v_accvgpr_write_b32 a14, v234

-- same size, no branch lands between them -- and written to <lib>_patched.so.  MI355X: the original fails the check, the patched library
passes it (profiles/r06_incident_i_patched.txt)."""
import sys
pat = bytes([0x14, 0x40, 0xd9, 0xd3, 0xeb, 0x01, 0x00, 0x18, 0x0e, 0x40, 0xd9, 0xd3, 0xea, 0x01, 0x00, 0x18, 0x3c, 0x01, 0xce, 0xbe, 0x7e, 0x00, 0xfe, 0x87])
new = pat[16:24] + pat[0:16]
name = sys.argv[1]
b = open(name, 'rb').read()
assert b.count(pat) == 1, 'expected one occurrence, found %d' % b.count(pat)
i = b.find(pat)
out = name.replace('.so', '_patched.so')
open(out, 'wb').write(b[:i] + new + b[i + len(pat):])
print('patched at offset %#x -> %s' % (i, out))
