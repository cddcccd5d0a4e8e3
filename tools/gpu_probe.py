import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/swift-qwen3-tts_amd')
from qwen3tts import synth, Qwen3TTSModel, GenerationRequest
from oracle import oracle as O
def bf(b): return (np.asarray(b,np.uint16).astype(np.uint32)<<16).view(np.float32)
name = sys.argv[1] if len(sys.argv)>1 else 'tiny-a'
d='/tmp/ckpt_'+name
synth.write_checkpoint(d, name)
t=time.time(); m=Qwen3TTSModel.from_pretrained(d, max_batch=4, max_frames=64, max_prompt=64, use_graph=False); print('load', time.time()-t, m.tts_model_type, m.supported_speakers)
om=O.OracleModel(d)
# linear
rng=np.random.default_rng(0)
x=synth.f32_to_bf16_bits(rng.standard_normal((5,256)).astype(np.float32)); W=synth.f32_to_bf16_bits((rng.standard_normal((48,256))*0.05).astype(np.float32))
y=m.debug_linear(x,W); 
import ctypes as C
yo=np.empty((5,48),np.uint16); O.lib().o_linear_bf16(O._p16(x),O._p16(W),None,5,256,48,O._p16(yo))
print('linear maxdiff', np.abs(bf(y)-bf(yo)).max(), 'mismatch', (y!=yo).mean())
pr=synth.synthetic_prompt(0,n_text=12,text_vocab=1000,im_start=1000,im_end=1001)
req=GenerationRequest(pr['text_ids'],12,None,'aiden','english')
oreq=O.Request(text_ids=pr['text_ids'],target_token_count=12,speaker='aiden',language='english')
ie,tr,pad=m.debug_prepare_inputs(req); oie,otr,opad=om.prepare_generation_inputs(oreq)
print('prompt shapes', ie.shape, oie.shape, tr.shape, otr.shape, 'diff', np.abs(bf(ie)-bf(oie)).max(), np.abs(bf(tr)-bf(otr)).max(), np.abs(bf(pad)-bf(opad[0])).max(), 'mism', (ie!=oie).mean())
F=6
otr_=om.generate_codes(oreq,O.Sampling(temperature=0.0,force_frames=F),keep_logits=True)
tl,cl,sampled=m.debug_generate_forced([req], otr_.codes[None], temperature=0.0)
otl=np.stack(otr_.talker_logits); ocl=np.stack(otr_.cp_logits)
print('talker logits diff', np.abs(bf(tl[0])-bf(otl)).max(axis=1), 'scale', np.abs(bf(otl)).max())
print('cp logits diff', np.abs(bf(cl[0])-bf(ocl)).max(axis=(1,2)))
print('sampled==oracle', (sampled[0]==otr_.codes).mean())
res=m.generate_batch([req], temperature=0.0, force_frames=F)
print('free-run codes equal', (res[0].codes==otr_.codes).mean(), res[0].audio.shape)
pcm,valid=om.codec_decode(otr_.codes)
pg,lens=m.codec_decode(otr_.codes)
print('pcm diff', np.abs(pg[0]-pcm).max(), lens, valid, 'clipfrac', (np.abs(pcm)>=1).mean())
for st in ('quantizer','pre_conv','pre_transformer','upsample0','upsample1','init_conv','block0','block1','block2','block3'):
    s={}; 
    a=m.debug_codec_stage(otr_.codes, st)
    om.codec_decode(otr_.codes, s)
    print(st, a.shape, 'maxdiff', np.abs(a-s[st]).max(), 'scale', np.abs(s[st]).max())
