import itertools
GROUPS = [list(range(0,4))+list(range(12,16))+list(range(20,28)),
          list(range(4,12))+list(range(16,20))+list(range(28,32)),
          list(range(32,36))+list(range(44,48))+list(range(52,60)),
          list(range(36,44))+list(range(48,52))+list(range(60,64))]
def cur_slot(row,kg,x): return 2*((kg + 2*((row>>3)&1))&3) + (x ^ ((row>>1)&1))
def read_ok(slot, stride_rows_128=True):
    # fragment read: lane -> (r16 = lane&15, kg = lane>>4), fixed x; address = r16*128 + slot*16 ; bank row 256 B
    for x in (0,1):
        for g in GROUPS:
            pos = set()
            for l in g:
                r, kg = l & 15, l >> 4
                pos.add(((r & 1) * 8 + slot(r, kg, x)))
            if len(pos) != 16: return False
    return True
def write_conf(slot):
    # ds_write_b128: 8 contiguous lanes (c16 = 8a..8a+7, fixed kg), rows stride 256 B (+ half select by row&1 -> 128 B = same bank mod 128)
    worst = 1
    for x in (0,1):
        for kg in range(4):
            for a in (0,1):
                from collections import Counter
                c = Counter(slot(8*a + i, kg, x) for i in range(8))
                worst = max(worst, max(c.values()))
    return worst
print("current: read_ok", read_ok(cur_slot), "write worst-way", write_conf(cur_slot))
# linear family: slot = (2*kg + x) ^ g(row), g = M * rowbits (3x4 over GF(2)); also allow kg rotation forms
best=[]
for M in itertools.product(range(8), repeat=4):   # M[b] = contribution (3 bits) of row bit b
    def g(row, M=M):
        v = 0
        for b in range(4):
            if (row >> b) & 1: v ^= M[b]
        return v
    s = lambda row,kg,x: ((2*kg + x) ^ g(row))
    if read_ok(s):
        best.append((write_conf(s), M))
best.sort()
print(len(best), best[:10])
