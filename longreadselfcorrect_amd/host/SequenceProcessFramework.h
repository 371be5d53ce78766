// SequenceProcessFramework.h -- the reference's generic dispatcher
// (Concurrency/SequenceProcessFramework.h:362-386) re-cut for a device back end.
//
// Same concepts, same guarantees:
//   Processor(const Parameter&);      Output Processor::process(const Input&)             -- classic, per item
//   PostProcessor(const Parameter&);  void PostProcessor::process(const Input&, const Output&)
//       -> called on the calling thread, once per input, in INPUT order, after the batch that
//          contains the item has been processed (reference: SequenceProcessFramework.h:183-195);
//          destroyed at the end of the run (its destructor prints the statistics, :385).
// New: a BATCHED processor concept for device back ends,
//   std::vector<Output> Processor::process_batch(const std::vector<Input>&)
// detected at compile time; a classic per-item Processor still plugs in unchanged.
// `thread` keeps its meaning for classic processors only in the sense of batch sizing
// (BUFFER_SIZE * thread items per batch, :26,155-195); a device processor shards inside process_batch.
#pragma once
#include <cstdio>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include <chrono>

#include "SequenceWorkItem.h"

namespace stride {
namespace SequenceProcessFramework {

const size_t BUFFER_SIZE = 500;        // reference :26

template <class P, class Input, class = void>
struct has_process_batch : std::false_type {};
template <class P, class Input>
struct has_process_batch<P, Input, decltype(void(std::declval<P&>().process_batch(std::declval<const std::vector<Input>&>())))>
    : std::true_type {};

template <class Input, class Output, class Processor>
typename std::enable_if<has_process_batch<Processor, Input>::value, std::vector<Output>>::type
run_batch(Processor& p, const std::vector<Input>& items)
{
    return p.process_batch(items);
}
template <class Input, class Output, class Processor>
typename std::enable_if<!has_process_batch<Processor, Input>::value, std::vector<Output>>::type
run_batch(Processor& p, const std::vector<Input>& items)
{
    std::vector<Output> out;
    out.reserve(items.size());
    for(const Input& it : items) out.push_back(p.process(it));
    return out;
}

// processSequences<Input, Output, Processor, PostProcessor, Parameter>(thread, readsFile, params)
template <class Input, class Output, class Processor, class PostProcessor, class Parameter>
size_t processSequences(int thread, const std::string& readsFile, const Parameter& params, size_t batch_items = 0)
{
    const auto t0 = std::chrono::steady_clock::now();
    SeqReader reader(readsFile);
    WorkItemGenerator<Input> generator(&reader);
    Processor processor(params);
    PostProcessor postProcessor(params);
    if(batch_items == 0) batch_items = BUFFER_SIZE * (size_t)(thread > 0 ? thread : 1);

    std::vector<Input> items;
    bool more = true;
    while(more) {
        items.clear();
        Input wi;
        while(items.size() < batch_items && (more = generator.generate(wi))) items.push_back(wi);
        if(items.empty()) break;
        const std::vector<Output> outs = run_batch<Input, Output, Processor>(processor, items);
        for(size_t i = 0; i < items.size(); ++i) postProcessor.process(items[i], outs[i]);
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::fprintf(stderr, "Processed %zu sequences (%lfs elapsed)\n", generator.getNumConsumed(), el);
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(stderr, "Processed %zu sequences in %lfs (%lf sequences/s)\n", generator.getNumConsumed(), secs,
                 (double)generator.getNumConsumed() / secs);
    return generator.getNumConsumed();
}

} // namespace SequenceProcessFramework
} // namespace stride
