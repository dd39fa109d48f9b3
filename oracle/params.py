"""Curve parameters -- data copied from the reference's parameter modules.

Sources (all under /root/reference/src/concrete/):
  bls12-377.params.ts:11-45, pasta.params.ts:10-46, bls12-381.params.ts:6-50,
  ed-on-bls12-377.params.ts:5-31.
Derived values (pallas lambda/beta) are recomputed with the same formulas the reference uses
(pasta.params.ts:19-32) rather than pasted.
"""

# ---------------------------------------------------------------- BLS12-377 G1
_p377 = 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001
_q377 = 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001

BLS12_377 = dict(
    label="bls12-377",
    kind="weierstrass",
    curve_id=0,
    modulus=_p377,
    order=_q377,
    cofactor=0x170B5D44300000000000000000000000,
    a=0,
    b=1,
    generator=dict(
        x=0x008848DEFE740A67C8FC6225BF87FF5485951E2CAA9D41BB188282C8BD37CB5CD5481512FFCD394EEAB9B16EB21BE9EF,
        y=0x01914A69C5102EFF1F674F5D30AFEEC4BD7FB348CA3E52D96D182AD44FB82305C2FE3D3634A9591AFD82DE55559C8EA6,
    ),
    endomorphism=dict(
        lambda_=0x12AB655E9A2CA55660B44D1E5C37B00114885F32400000000000000000000000,
        beta=0x1AE3A4617C510EABC8756BA8F8C524EB8882A75CC9BC8E359064EE822FB5BFFD1E945779FFFFFFFFFFFFFFFFFFFFFFF,
    ),
    fe_bytes=48,
)

# ---------------------------------------------------------------- Pallas
_pP = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
_qP = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
_lamP = pow(5, (_qP - 1) // 3, _qP)  # pasta.params.ts:20
_beta2P = pow(5, (_pP - 1) // 3, _pP)  # pasta.params.ts:29
_betaP = _beta2P * _beta2P % _pP  # pasta.params.ts:30
assert pow(_lamP, 3, _qP) == 1 and pow(_betaP, 3, _pP) == 1

PALLAS = dict(
    label="pallas",
    kind="weierstrass",
    curve_id=1,
    modulus=_pP,
    order=_qP,
    cofactor=1,
    a=0,
    b=5,
    generator=dict(
        x=1,
        y=0x1B74B5A30A12937C53DFA9F06378EE548F655BD4333D477119CF7A23CAED2ABB,
    ),
    endomorphism=dict(lambda_=_lamP, beta=_betaP),
    fe_bytes=32,
)

# ---------------------------------------------------------------- BLS12-381 G1
_p381 = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
_q381 = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
_minusZ = 0xD201000000010000

BLS12_381 = dict(
    label="bls12-381",
    kind="weierstrass",
    curve_id=2,
    modulus=_p381,
    order=_q381,
    cofactor=0x396C8C005555E1568C00AAAB0000AAAB,
    a=0,
    b=4,
    generator=dict(
        x=0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
        y=0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
    ),
    endomorphism=dict(
        lambda_=_minusZ**2 - 1,  # bls12-381.params.ts:24 (lambda2)
        beta=0x1A0111EA397FE699EC02408663D4DE85AA0D857D89759AD4897D29650FB85F9B409427EB4F49FFFD8BFD00000000AAAC,
    ),
    fe_bytes=48,
)

# ---------------------------------------------------------------- ed-on-bls12-377 (twisted Edwards, a = -1)
ED_ON_BLS12_377 = dict(
    label="ed-on-bls12-377",
    kind="twisted-edwards",
    curve_id=3,
    modulus=_q377,
    order=0x4AAD957A68B2955982D1347970DEC005293A3AFC43C8AFEB95AEE9AC33FD9FF,
    cofactor=4,
    d=3021,
    generator=dict(
        x=0x9F1B5A5BAF6ACF06FED91C9AE9EBFA06068DD2835790980894E2328F3EBCA05,
        y=0x9A20DF36571AC3CD906B256080BA8454453C177AAF3131BB50A67BF1A806781,
    ),
    fe_bytes=32,
)

CURVES = {c["label"]: c for c in (BLS12_377, PALLAS, BLS12_381, ED_ON_BLS12_377)}
CURVE_BY_ID = {c["curve_id"]: c for c in CURVES.values()}

# Known-answer points held by the reference's own smoke tests.
# scripts/zprize23/submission-test-bls377.ts:6-10
KAT_BLS12_377_POINT = dict(
    x=111871295567327857271108656266735188604298176728428155068227918632083036401841336689521497731900230387779623820740,
    y=76860045326390600098227152997486448974650822224305058012700629806287380625419427989664237630603922765089083164740,
)
# scripts/zprize23/submission-test.ts:5-10
KAT_ED377_POINT = dict(
    x=2796670805570508460920584878396618987767121022598342527208237783066948667246,
    y=8134280397689638111748378379571739274369602049665521098046934931245960532166,
    t=3446088593515175914550487355059397868296219355049460558182099906777968652023,
)
