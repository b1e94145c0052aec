"""Build-time checks that need hipcc but no GPU."""


def test_wgrad_instances_stay_inside_their_register_budget():
    """conv_wgrad.hip waits by hand for inline-asm loads issued one or two tiles earlier: that is only sound while the
    register allocator never has to park a value (AGPR copy / scratch spill), i.e. while every instance needs clearly fewer
    than 256 architectural VGPRs and no scratch (segmentation_amd/_wgrad_regs.py; DESIGN.md 'register budget')."""
    from segmentation_amd import _wgrad_regs as chk
    inst = chk.instances()
    assert len(inst) >= 30
    bad = [i for i in inst if i['arch_vgprs'] >= 250 or i['spills'] or i['scratch_bytes']]
    assert not bad, bad
