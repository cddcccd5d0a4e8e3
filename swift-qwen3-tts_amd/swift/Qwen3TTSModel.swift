//
//  Qwen3TTSModel.swift -- Swift shim over the C ABI in include/q3tts.h.
//
//  Keeps the public surface of AtomGradient/swift-qwen3-tts (Sources/Qwen3TTS/Models/Qwen3.swift:30,
//  :1291-1301, :1382; Qwen3+Streaming.swift:8-18; Core/GenerationTypes.swift:15-84) while the MLX backend is
//  replaced by libq3tts_hip.so. NOT compiled in this repository (no Swift toolchain in the build image);
//  it is the binding a maintainer adds next to a `module.modulemap` exposing q3tts.h as `CQ3TTS`:
//
//      module CQ3TTS { header "q3tts.h"  link "q3tts_hip"  export * }
//
//  Differences a caller sees: audio comes back as [Float] (MLX is gone), a `seed:` argument exists, and two things are new:
//  `generateBatch` (rows are independent: each equals the single-request result) and `generateChunkedStream` (the waveform in
//  pieces, optionally while tokens are still being generated). Tokenisation stays in Swift via swift-transformers exactly as
//  in the reference. `config` keeps its type: Config.swift is plain Codable data (its one import, MLXLMCommon, goes with the
//  MLX targets) and the shim decodes the same config.json the reference decodes (Qwen3.swift:1386-1388).
//
import CQ3TTS
import Foundation
import Tokenizers

public enum AudioGenerationError: Error, LocalizedError {   // GenerationTypes.swift:63-84
    case modelNotInitialized(String), generationFailed(String), invalidInput(String)
    case audioDecodingFailed(String), audioEncodingFailed(String)
    public var errorDescription: String? {
        switch self {   // the engine already returns the reference's full message text
        case .modelNotInitialized(let m), .generationFailed(let m), .invalidInput(let m),
             .audioDecodingFailed(let m), .audioEncodingFailed(let m): return m
        }
    }
    static func from(_ st: q3tts_status, _ msg: String) -> AudioGenerationError {
        switch st {
        case Q3TTS_ERR_MODEL_NOT_INITIALIZED: return .modelNotInitialized(msg)
        case Q3TTS_ERR_GENERATION_FAILED: return .generationFailed(msg)
        case Q3TTS_ERR_INVALID_INPUT: return .invalidInput(msg)
        case Q3TTS_ERR_AUDIO_DECODING_FAILED: return .audioDecodingFailed(msg)
        case Q3TTS_ERR_AUDIO_ENCODING_FAILED: return .audioEncodingFailed(msg)
        default: return .modelNotInitialized(msg)   // IO / device failures surface at load time
        }
    }
}
public typealias Qwen3TTSError = AudioGenerationError

public struct AudioGenerationInfo: Sendable {   // GenerationTypes.swift:15-21
    public let promptTokenCount: Int, generationTokenCount: Int
    public let prefillTime: TimeInterval, generateTime: TimeInterval
    public let tokensPerSecond: Double, peakMemoryUsage: Double

    /// GenerationTypes.swift:39-45 (same three lines, same number formats)
    public var summary: String {
        let promptRate = String(format: "%.2f", Double(promptTokenCount) / max(prefillTime, 0.001))
        return """
        Prompt:     \(promptTokenCount) tokens, \(promptRate) tokens/s, \(String(format: "%.3f", prefillTime))s
        Generation: \(generationTokenCount) tokens, \(String(format: "%.2f", tokensPerSecond)) tokens/s, \(String(format: "%.3f", generateTime))s
        Peak Memory Usage: \(peakMemoryUsage) GB
        """
    }
}

public enum AudioGeneration: Sendable {   // GenerationTypes.swift:51-58
    case token(Int)
    case info(AudioGenerationInfo)
    case audio([Float])
}
public typealias Qwen3TTSGeneration = AudioGeneration

/// One row of `generateBatch`: the arguments of `generate` that differ per utterance.
public struct Qwen3TTSBatchRequest: Sendable {
    public var text: String, speaker: String?, instruct: String?, language: String
    public init(text: String, speaker: String? = nil, instruct: String? = nil, language: String = "auto") {
        self.text = text; self.speaker = speaker; self.instruct = instruct; self.language = language
    }
}

/// Events of `generateChunkedStream`: the reference's three cases (Core/GenerationTypes.swift:51-58) plus the waveform in pieces.
/// A separate enum, so that exhaustive switches over the reference's `AudioGeneration` keep compiling.
public enum Qwen3TTSChunkedGeneration: Sendable {
    case token(Int)
    case audioChunk(offset: Int, samples: [Float])   // samples [offset, offset + count) of the final audio, in order
    case info(AudioGenerationInfo)
    case audio([Float])
}

public final class Qwen3TTSModel {
    private var handle: OpaquePointer?
    public var tokenizer: Tokenizer?
    private let info: q3tts_model_info
    /// Qwen3.swift:31 -- the decoded config.json (Config.swift's type, unchanged)
    public let config: Qwen3TTSModelConfig

    private init(handle: OpaquePointer, info: q3tts_model_info, config: Qwen3TTSModelConfig) {
        self.handle = handle; self.info = info; self.config = config
    }
    deinit { q3tts_model_free(handle) }

    /// fromPretrained(_:) -- Qwen3.swift:1382
    public static func fromPretrained(_ modelPath: String, device: Int32 = 0, maxBatch: Int32 = 1) async throws -> Qwen3TTSModel {
        // Qwen3.swift:1386-1388: the same decode of the same file (a malformed config fails here, before the engine sees it)
        let configData = try Data(contentsOf: URL(fileURLWithPath: modelPath).appendingPathComponent("config.json"))
        let config = try JSONDecoder().decode(Qwen3TTSModelConfig.self, from: configData)
        var opts = q3tts_load_opts()
        q3tts_default_load_opts(&opts)
        opts.device = device
        opts.max_batch = maxBatch
        var h: OpaquePointer?
        let st = q3tts_model_load(modelPath, &opts, &h)
        guard st == Q3TTS_OK, let h else { throw AudioGenerationError.from(st, String(cString: q3tts_last_error(nil))) }
        var mi = q3tts_model_info()
        q3tts_model_get_info(h, &mi)
        let m = Qwen3TTSModel(handle: h, info: mi, config: config)
        m.tokenizer = try await AutoTokenizer.from(modelFolder: URL(fileURLWithPath: modelPath))   // Qwen3.swift:1458
        return m
    }

    public var sampleRate: Int { Int(info.sample_rate) }                                    // Qwen3.swift:1262
    public var ttsModelType: String { withUnsafeBytes(of: info.tts_model_type) { String(cString: $0.bindMemory(to: CChar.self).baseAddress!) } }
    public var supportsVoiceCloning: Bool { info.supports_voice_cloning != 0 }                // Qwen3.swift:1210
    public var hasVoiceCloning: Bool { info.has_voice_cloning != 0 }                          // Qwen3.swift:61
    public var supportedSpeakers: [String] {                                                 // Qwen3.swift:965
        (0..<q3tts_model_num_speakers(handle)).map { String(cString: q3tts_model_speaker_name(handle, $0)) }
    }

    /// generate(text:speaker:instruct:language:temperature:topK:topP:repetitionPenalty:maxTokens:) -- Qwen3.swift:1291-1301
    public func generate(text: String, speaker: String? = nil, instruct: String? = nil, language: String = "auto",
                         temperature: Float = 0.9, topK: Int = 50, topP: Float = 1.0, repetitionPenalty: Float = 1.05,
                         maxTokens: Int = 2048, seed: UInt64 = 0) async throws -> [Float] {
        try run(text: text, speaker: speaker, instruct: instruct, language: language, temperature: temperature, topK: topK,
                topP: topP, repetitionPenalty: repetitionPenalty, maxTokens: maxTokens, seed: seed, onEvent: nil)
    }

    /// generateVoiceDesign(text:language:instruct:...:onToken:) -- Qwen3.swift:587-597. `onToken` receives every first-codebook
    /// id as it is generated (:698), EOS excluded. `route: 1` asks the engine for this prompt builder whatever the checkpoint's
    /// tts_model_type, as the reference's direct call does.
    public func generateVoiceDesign(text: String, language: String = "auto", instruct: String? = nil, temperature: Float = 0.9,
                                    topK: Int = 50, topP: Float = 1.0, repetitionPenalty: Float = 1.05, maxTokens: Int = 2048,
                                    seed: UInt64 = 0, onToken: ((Int) -> Void)? = nil) throws -> [Float] {
        try run(text: text, speaker: nil, instruct: instruct, language: language, temperature: temperature, topK: topK,
                topP: topP, repetitionPenalty: repetitionPenalty, maxTokens: maxTokens, seed: seed, route: 1,
                onEvent: onToken.map { cb in { ev in if case .token(let t) = ev { cb(t) } } })
    }

    /// generateCustomVoice(text:speaker:language:instruct:...:onToken:) -- Qwen3.swift:783-794; the speaker is validated against
    /// talker_config.spk_id with the reference's message (:803-811).
    public func generateCustomVoice(text: String, speaker: String, language: String = "auto", instruct: String? = nil,
                                    temperature: Float = 0.9, topK: Int = 50, topP: Float = 1.0, repetitionPenalty: Float = 1.05,
                                    maxTokens: Int = 2048, seed: UInt64 = 0, onToken: ((Int) -> Void)? = nil) throws -> [Float] {
        try run(text: text, speaker: speaker, instruct: instruct, language: language, temperature: temperature, topK: topK,
                topP: topP, repetitionPenalty: repetitionPenalty, maxTokens: maxTokens, seed: seed, route: 2,
                onEvent: onToken.map { cb in { ev in if case .token(let t) = ev { cb(t) } } })
    }

    /// generateStream(...) -- Qwen3+Streaming.swift:8-18: .token per frame, then .info, then .audio
    public func generateStream(text: String, speaker: String? = nil, instruct: String? = nil, language: String = "auto",
                               temperature: Float = 0.9, topK: Int = 50, topP: Float = 1.0, repetitionPenalty: Float = 1.05,
                               maxTokens: Int = 2048, seed: UInt64 = 0) -> AsyncThrowingStream<Qwen3TTSGeneration, Error> {
        AsyncThrowingStream { continuation in
            Thread.detachNewThread {   // same threading model as the reference (Qwen3+Streaming.swift:19-20)
                do {
                    _ = try self.run(text: text, speaker: speaker, instruct: instruct, language: language,
                                     temperature: temperature, topK: topK, topP: topP, repetitionPenalty: repetitionPenalty,
                                     maxTokens: maxTokens, seed: seed) { continuation.yield($0) }
                    continuation.finish()
                } catch { continuation.finish(throwing: error) }
            }
        }
    }

    /// New (the reference decodes one-shot, README.md:140): `generateStream` with the waveform in pieces of `chunkFrames` codec
    /// frames (80 ms each). windowFrames == 0: the pieces are cut from the exact decode after the last token (bit-identical to
    /// `.audio`); windowFrames > 0: they leave WHILE tokens are still being generated (include/q3tts.h `audio_window_frames`).
    public func generateChunkedStream(text: String, speaker: String? = nil, instruct: String? = nil, language: String = "auto",
                                      temperature: Float = 0.9, topK: Int = 50, topP: Float = 1.0, repetitionPenalty: Float = 1.05,
                                      maxTokens: Int = 2048, seed: UInt64 = 0, chunkFrames: Int = 16, windowFrames: Int = 32)
        -> AsyncThrowingStream<Qwen3TTSChunkedGeneration, Error> {
        AsyncThrowingStream { continuation in
            Thread.detachNewThread {
                do {
                    _ = try self.run(text: text, speaker: speaker, instruct: instruct, language: language,
                                     temperature: temperature, topK: topK, topP: topP, repetitionPenalty: repetitionPenalty,
                                     maxTokens: maxTokens, seed: seed, chunkFrames: chunkFrames, windowFrames: windowFrames,
                                     onChunk: { off, pcm in continuation.yield(.audioChunk(offset: off, samples: pcm)) }) { ev in
                        switch ev {
                        case .token(let t): continuation.yield(.token(t))
                        case .info(let i): continuation.yield(.info(i))
                        case .audio(let a): continuation.yield(.audio(a))
                        }
                    }
                    continuation.finish()
                } catch { continuation.finish(throwing: error) }
            }
        }
    }

    /// New: several utterances in one call (up to the `maxBatch` the model was loaded with). Rows are independent -- each
    /// result equals what `generate` returns for that request with the same seed -- and a row that fails (no tokens) is nil.
    public func generateBatch(_ requests: [Qwen3TTSBatchRequest], temperature: Float = 0.9, topK: Int = 50, topP: Float = 1.0,
                              repetitionPenalty: Float = 1.05, maxTokens: Int = 2048, seed: UInt64 = 0) throws -> [[Float]?] {
        guard let tokenizer else { throw AudioGenerationError.modelNotInitialized("Model not initialized: Tokenizer not loaded") }
        let n = requests.count
        guard n > 0 else { return [] }
        // everything a q3tts_request points at lives in these arrays until the call returns
        let textIds = requests.map { tokenizer.encode(text: "<|im_start|>assistant\n\($0.text)<|im_end|>\n<|im_start|>assistant\n").map(Int32.init) }
        let instructIds = requests.map { r in (r.instruct?.isEmpty == false)
            ? tokenizer.encode(text: "<|im_start|>user\n\(r.instruct!)<|im_end|>\n").map(Int32.init) : [Int32]() }
        let speakers = requests.map { $0.speaker.map { strdup($0) } ?? nil }
        let languages = requests.map { strdup($0.language) }
        defer { speakers.forEach { free($0) }; languages.forEach { free($0) } }
        let textBufs = textIds.map { ids -> UnsafeMutablePointer<Int32> in
            let p = UnsafeMutablePointer<Int32>.allocate(capacity: max(ids.count, 1)); p.initialize(from: ids, count: ids.count); return p }
        let instBufs = instructIds.map { ids -> UnsafeMutablePointer<Int32> in
            let p = UnsafeMutablePointer<Int32>.allocate(capacity: max(ids.count, 1)); p.initialize(from: ids, count: ids.count); return p }
        defer { textBufs.forEach { $0.deallocate() }; instBufs.forEach { $0.deallocate() } }
        var reqs = [q3tts_request](repeating: q3tts_request(), count: n)
        for i in 0..<n {
            reqs[i].text_ids = UnsafePointer(textBufs[i]); reqs[i].n_text_ids = Int32(textIds[i].count)
            reqs[i].instruct_ids = instructIds[i].isEmpty ? nil : UnsafePointer(instBufs[i]); reqs[i].n_instruct_ids = Int32(instructIds[i].count)
            reqs[i].target_token_count = Int32(tokenizer.encode(text: requests[i].text).count)
            reqs[i].speaker = UnsafePointer(speakers[i]); reqs[i].language = UnsafePointer(languages[i])
            reqs[i].max_tokens = Int32(maxTokens)
        }
        var sampling = q3tts_sampling()
        q3tts_default_sampling(&sampling)
        sampling.temperature = temperature; sampling.top_k = Int32(topK); sampling.top_p = topP
        sampling.repetition_penalty = repetitionPenalty; sampling.seed = seed
        var results = [q3tts_result](repeating: q3tts_result(), count: n)
        defer { q3tts_result_free(&results, Int32(n)) }
        let st = q3tts_generate(handle, &reqs, Int32(n), &sampling, nil, nil, &results)
        guard st == Q3TTS_OK else { throw AudioGenerationError.from(st, String(cString: q3tts_last_error(handle))) }
        return results.map { $0.status == Q3TTS_OK ? Array(UnsafeBufferPointer(start: $0.pcm, count: Int($0.n_samples))) : nil }
    }

    /// generateVoiceClone(text:referenceAudio:referenceText:language:...) -- Qwen3.swift:1009-1020 (repetition penalty 1.5)
    public func generateVoiceClone(text: String, referenceAudio: [Float], referenceText: String, language: String = "auto",
                                   temperature: Float = 0.9, topK: Int = 50, topP: Float = 1.0, repetitionPenalty: Float = 1.5,
                                   maxTokens: Int = 2048, seed: UInt64 = 0, onToken: ((Int) -> Void)? = nil) throws -> [Float] {
        guard let tokenizer else { throw AudioGenerationError.modelNotInitialized("Model not initialized: Tokenizer not loaded") }
        // Qwen3.swift:448-449: the reference transcript in the assistant template
        let refIds = tokenizer.encode(text: "<|im_start|>assistant\n\(referenceText)<|im_end|>\n").map(Int32.init)
        return try run(text: text, speaker: nil, instruct: nil, language: language, temperature: temperature, topK: topK,
                       topP: topP, repetitionPenalty: repetitionPenalty, maxTokens: maxTokens, seed: seed,
                       referenceAudio: referenceAudio, refTextIds: refIds,
                       onEvent: onToken.map { cb in { ev in if case .token(let t) = ev { cb(t) } } })
    }

    /// extractSpeakerEmbedding(_:sampleRate:) -- Qwen3.swift:222-249
    public func extractSpeakerEmbedding(_ audio: [Float], sampleRate: Int = 24000) throws -> [Float] {
        var out = [Float](repeating: 0, count: Int(info.speaker_embedding_dim))
        let st = audio.withUnsafeBufferPointer { a in
            q3tts_speaker_embedding(handle, a.baseAddress, Int64(a.count), Int32(sampleRate), &out, Int32(out.count))
        }
        guard st == Q3TTS_OK else { throw AudioGenerationError.from(st, String(cString: q3tts_last_error(handle))) }
        return out
    }

    private func run(text: String, speaker: String?, instruct: String?, language: String, temperature: Float, topK: Int,
                     topP: Float, repetitionPenalty: Float, maxTokens: Int, seed: UInt64, route: Int32 = 0,
                     referenceAudio: [Float] = [], refTextIds: [Int32] = [], chunkFrames: Int = 0, windowFrames: Int = 0,
                     onChunk: ((Int, [Float]) -> Void)? = nil,
                     onEvent: ((Qwen3TTSGeneration) -> Void)?) throws -> [Float] {
        guard let tokenizer else { throw AudioGenerationError.modelNotInitialized("Model not initialized: Tokenizer not loaded") }
        // the three tokenisations of the reference (Qwen3.swift:274-275, 364-365, 822)
        let textIds = tokenizer.encode(text: "<|im_start|>assistant\n\(text)<|im_end|>\n<|im_start|>assistant\n").map(Int32.init)
        let instructIds = (instruct?.isEmpty == false)
            ? tokenizer.encode(text: "<|im_start|>user\n\(instruct!)<|im_end|>\n").map(Int32.init) : []
        let targetCount = Int32(tokenizer.encode(text: text).count)
        // defaults first, then the fields this call sets: fields added to the C struct later keep their defaults here
        var sampling = q3tts_sampling()
        q3tts_default_sampling(&sampling)
        sampling.temperature = temperature
        sampling.top_k = Int32(topK)
        sampling.top_p = topP
        sampling.repetition_penalty = repetitionPenalty
        sampling.seed = seed
        sampling.audio_chunk_frames = Int32(chunkFrames)      // 0: one-shot like the reference
        sampling.audio_window_frames = Int32(windowFrames)    // > 0: chunks leave while tokens are still being generated
        var result = q3tts_result()
        let box = Unmanaged.passRetained(EventBox(onEvent, onChunk))
        defer { box.release(); q3tts_result_free(&result, 1) }
        let st: q3tts_status = textIds.withUnsafeBufferPointer { tp in
            instructIds.withUnsafeBufferPointer { ip in
                referenceAudio.withUnsafeBufferPointer { ra in
                    refTextIds.withUnsafeBufferPointer { rt in
                        withOptionalCString(speaker) { sp in
                            language.withCString { lp in
                                // zero-initialised, then field by field (like q3tts_sampling above): a field added to the C
                                // struct later keeps its zero default here instead of breaking a memberwise initialiser
                                var req = q3tts_request()
                                req.text_ids = tp.baseAddress; req.n_text_ids = Int32(tp.count)
                                req.instruct_ids = ip.count > 0 ? ip.baseAddress : nil; req.n_instruct_ids = Int32(ip.count)
                                req.target_token_count = targetCount
                                req.speaker = sp; req.language = lp
                                req.max_tokens = Int32(maxTokens)
                                req.ref_audio = ra.count > 0 ? ra.baseAddress : nil; req.n_ref_samples = Int64(ra.count)   // generateVoiceClone
                                req.ref_text_ids = rt.count > 0 ? rt.baseAddress : nil; req.n_ref_text_ids = Int32(rt.count)
                                req.route = route   // 1 / 2: generateVoiceDesign / generateCustomVoice called directly
                                return q3tts_generate(handle, &req, 1, &sampling, (onEvent == nil && onChunk == nil) ? nil : eventTrampoline, box.toOpaque(), &result)
                            }
                        }
                    }
                }
            }
        }
        guard st == Q3TTS_OK else { throw AudioGenerationError.from(st, String(cString: q3tts_last_error(handle))) }
        guard result.status == Q3TTS_OK else { throw AudioGenerationError.generationFailed("Generation failed: No tokens generated") }
        return Array(UnsafeBufferPointer(start: result.pcm, count: Int(result.n_samples)))
    }
}

private final class EventBox {
    let f: ((Qwen3TTSGeneration) -> Void)?
    let chunk: ((Int, [Float]) -> Void)?
    init(_ f: ((Qwen3TTSGeneration) -> Void)?, _ chunk: ((Int, [Float]) -> Void)? = nil) { self.f = f; self.chunk = chunk }
}

private let eventTrampoline: q3tts_event_cb = { user, evp in
    guard let user, let ev = evp?.pointee else { return }
    let box = Unmanaged<EventBox>.fromOpaque(user).takeUnretainedValue()
    switch ev.kind {
    case Q3TTS_EVENT_TOKEN: box.f?(.token(Int(ev.token)))
    case Q3TTS_EVENT_INFO:
        let i = ev.info.pointee
        box.f?(.info(AudioGenerationInfo(promptTokenCount: Int(i.prompt_token_count), generationTokenCount: Int(i.generation_token_count),
                                         prefillTime: i.prefill_time, generateTime: i.generate_time,
                                         tokensPerSecond: i.tokens_per_second, peakMemoryUsage: i.peak_memory_usage)))
    case Q3TTS_EVENT_AUDIO: box.f?(.audio(Array(UnsafeBufferPointer(start: ev.pcm, count: Int(ev.n_samples)))))
    case Q3TTS_EVENT_AUDIO_CHUNK:   // only with audio_chunk_frames > 0 (generateChunkedStream): not part of the reference's enum
        box.chunk?(Int(ev.sample_offset), Array(UnsafeBufferPointer(start: ev.pcm, count: Int(ev.n_samples))))
    default: break
    }
}

private func withOptionalCString<R>(_ s: String?, _ body: (UnsafePointer<CChar>?) -> R) -> R {
    if let s { return s.withCString { body($0) } }
    return body(nil)
}
